"""A/B of two builds of the library on ONE GPU box (boxes differ by a few per cent between gpurun calls):
    python tools/ab.py <libA.so> <libB.so> [rounds]
Runs bench.py's default pass alternately (A B A B ...) and prints ms per frame and the dominant kernel's launch time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, ARAH_LIB_PATH=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-gpu-baseline", "--passes", "default",
                              "--no-train", "--steps", "8", "--warmup", "2"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[l].append((d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
        except Exception as e:   # noqa
            print("failed", l, out.stderr[-400:])
for l in libs:
    print(os.path.basename(l), " ".join("%.2f/%.2f" % x for x in res[l]))
