"""Launches of one training step's backward by autograd node AND input shapes (which ViewBackward / AddmmBackward / ... they are).
    python tools/probes/train_bwd_nodes.py [substring of the node name ...]"""
import os, sys, collections, bisect, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__; __graft_entry__.build()
from arah_release_amd import config, synthetic, training
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
model, cfg = config.build_synthetic_model("zju313", device=dev); model.train()
opt = training.configure_optimizers(model, cfg); crit = training.build_loss(cfg)
scene = synthetic.SyntheticScene(0)
batches = [scene.make_inputs(512, 512, frame_idx=k, max_rays=2048, eval_mode=False, device=dev) for k in range(6)]
def step(inp):
    opt.zero_grad(set_to_none=True)
    training.training_step(model, crit, inp)["loss"].backward()
    opt.step()
for k in range(4): step(batches[k])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(batches[4]); torch.cuda.synchronize()
ev = prof.events()
launches = sorted(e.time_range.start for e in ev if "LaunchKernel" in e.name or e.name in ("hipMemcpyAsync", "hipMemsetAsync"))
want = sys.argv[1:]
rows = collections.Counter()
for e in ev:
    if not e.name.startswith("autograd::engine::evaluate_function: "):
        continue
    name = e.name.split(": ", 1)[1]
    if want and not any(w in name for w in want):
        continue
    n = bisect.bisect_right(launches, e.time_range.end) - bisect.bisect_left(launches, e.time_range.start)
    if n:
        rows[(name, str(e.input_shapes)[:110])] += n
for (name, shp), n in rows.most_common(60):
    print("%4d  %-34s %s" % (n, name[:34], shp))
