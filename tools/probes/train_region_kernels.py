import os, sys, collections, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__; __graft_entry__.build()
from arah_release_amd import config, synthetic, training, renderer, hip
from torch.profiler import profile, ProfilerActivity, record_function
dev = torch.device("cuda", 0)
model, cfg = config.build_synthetic_model("zju313", device=dev); model.train()
opt = training.configure_optimizers(model, cfg); crit = training.build_loss(cfg)
scene = synthetic.SyntheticScene(0)
batches = [scene.make_inputs(512, 512, frame_idx=k, max_rays=2048, eval_mode=False, device=dev) for k in range(6)]
def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def g(*a, **k):
        with record_function("R:" + name):
            return f(*a, **k)
    setattr(obj, attr, g)
for name in sys.argv[1:]:
    mod, attr = name.rsplit(".", 1)
    wrap({"training": training, "renderer": renderer, "hip": hip, "sdf_decoder": model.sdf_decoder, "crit": crit, "idhr": model.idhr_network, "model": model}[mod], attr, name)
def step(inp):
    opt.zero_grad(set_to_none=True)
    training.training_step(model, crit, inp)["loss"].backward()
    opt.step()
for k in range(4): step(batches[k])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(batches[4]); torch.cuda.synchronize()
ev = prof.events()
ranges = [(e.name[2:], e.time_range.start, e.time_range.end) for e in ev if e.name.startswith("R:")]
# map launches to kernels by correlation: use cpu-side launch events and the aten op enclosing them
ops = sorted([(e.time_range.start, e.time_range.end, e.name) for e in ev if e.device_type == torch.autograd.DeviceType.CPU and not e.name.startswith("R:") and not e.name.startswith("hip")], key=lambda t: t[0])
launches = [e for e in ev if "LaunchKernel" in e.name or e.name in ("hipMemcpyAsync", "hipMemsetAsync", "hipMemcpyWithStream")]
for rname, a, b in ranges:
    cnt = collections.Counter()
    for e in launches:
        s = e.time_range.start
        if a <= s <= b:
            # innermost enclosing op
            best = None
            for (oa, ob, on) in ops:
                if oa <= s <= ob and (best is None or ob - oa < best[0]):
                    best = (ob - oa, on)
            cnt[best[1] if best else "(none)"] += 1
    print("== %s: %d launches" % (rname, sum(cnt.values())))
    for k, v in cnt.most_common(40): print("   %4d  %s" % (v, k[:90]))
