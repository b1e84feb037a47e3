import os, sys, time, faulthandler
sys.path.insert(0, os.getcwd())
faulthandler.dump_traceback_later(40, exit=True)
import torch
from arah_release_amd import config, renderer, synthetic
dev = torch.device("cuda:0")
scene = synthetic.SyntheticScene(0)
model, cfg = config.build_synthetic_model("zju377_mono", device=dev)
sizes = [int(x) for x in sys.argv[1].split(",")]
ns = int(sys.argv[2])
pre = sys.argv[3].startswith("pre")
nosync = sys.argv[3] == "prenosync"
fidx = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [3 * k for k in range(len(sizes))]
frames = [scene.make_inputs(s, s, frame_idx=f, device=dev) for f, s in zip(fidx, sizes)]
if pre:
    with torch.no_grad():
        for f in frames:
            model(dict(f), eval=True)
    if not nosync:
        torch.cuda.synchronize()
    print("sequential pass done", flush=True)
t0 = time.time()
outs = renderer.render_sequence(model, [dict(f) for f in frames], n_streams=ns, eval=True)
print("enqueued in %.3f s" % (time.time() - t0), flush=True)
torch.cuda.synchronize()
print("sizes %s streams %d pre %s: done in %.3f s" % (sizes, ns, pre, time.time() - t0), flush=True)
