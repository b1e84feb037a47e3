"""Host (enqueue) time of one eval frame against its GPU time: frames enqueued back to back on one stream without waiting."""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import __graft_entry__
__graft_entry__.build()
from arah_release_amd import config, synthetic, renderer
dev = torch.device("cuda", 0)
model, cfg = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
scene = synthetic.SyntheticScene(0)
frames = [scene.make_inputs(512, 512, frame_idx=k, device=dev) for k in range(12)]
with torch.no_grad():
    for f in frames[:3]:
        model(dict(f), eval=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); c0 = time.process_time()
    for f in frames[2:]:
        model(dict(f), eval=True)
    t1 = time.perf_counter(); c1 = time.process_time()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
n = len(frames) - 2
print("enqueue wall %.2f ms/frame, process CPU %.2f ms/frame, until the GPU is done %.2f ms/frame" % (1e3 * (t1 - t0) / n, 1e3 * (c1 - c0) / n, 1e3 * (t2 - t0) / n))
import cProfile, pstats
pr = cProfile.Profile()
with torch.no_grad():
    pr.enable()
    for f in frames[2:]:
        model(dict(f), eval=True)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
