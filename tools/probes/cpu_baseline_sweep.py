"""bench.py's multi-process CPU baseline for several (processes, threads) splits of the host's cores."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from arah_release_amd import synthetic

if __name__ == "__main__":
    scene = synthetic.SyntheticScene(0)
    rays = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    for spec in sys.argv[2:]:
        w, t = (int(x) for x in spec.split("x"))
        os.environ["ARAH_CPU_BASELINE_WORKERS"] = str(w)
        out = bench.cpu_baseline_multiprocess(scene, "zju377_mono", 512, 64, 16, 16, rays, None, None, threads_per_worker=t)
        print(spec, "%.0f rays/s" % out["value"], out["sample"][-60:], flush=True)
