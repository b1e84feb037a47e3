"""Probe: frames of a sequence alternating over S HIP streams (one workspace each) vs one stream."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from arah_release_amd import config, hip, synthetic

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
scene = synthetic.SyntheticScene(0)
model, cfg = config.build_synthetic_model("zju377_mono", device=dev)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NF = 12
inputs = [scene.make_inputs(512, 512, frame_idx=f, device=dev) for f in range(NF + 2)]
tracer = model.idhr_network.ray_tracer
streams = [torch.cuda.Stream(dev) for _ in range(S)]
wss = [hip.Workspace(dev) for _ in range(S)]
n_max = max(int(i["ray_dirs"].shape[1]) for i in inputs)
for w in wss:
    w.ensure(n_max, 64)
torch.cuda.synchronize()


def run_seq(multi):
    from arah_release_amd import renderer
    renderer.render_sequence(model, inputs[:2], n_streams=S if multi else 1, eval=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [o["rgb_values"] for o in renderer.render_sequence(model, inputs[2:], n_streams=S if multi else 1, eval=True)]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / NF, outs


def _old(multi):
    outs = [None] * len(inputs)
    with torch.no_grad():
        for k, inp in enumerate(inputs):
            if k == 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            pass
        torch.cuda.synchronize()
    return 0, outs


for rep in range(2):
    a, oa = run_seq(False)
    b, ob = run_seq(True)
    print("grid cap %s: one stream %.2f ms/frame, %d streams %.2f ms/frame, identical %s" % (
        os.environ.get("ARAH_MAX_GRID", "512"), 1e3 * a, S, 1e3 * b, all(torch.equal(x, y) for x, y in zip(oa, ob))))
