import torch, time
x = (torch.rand(2048, 64, device="cuda") > 0.1)
try:
    r = torch.nonzero_static(x, size=int(x.sum()))
    print("nonzero_static ok", r.shape, torch.equal(r, x.nonzero()))
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        r = torch.nonzero_static(x, size=r.shape[0]); torch.cuda.synchronize()
    print([e.name for e in prof.events() if "Launch" in e.name or "Memcpy" in e.name or "Synchronize" in e.name])
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        r = x.nonzero(); torch.cuda.synchronize()
    print([e.name for e in prof.events() if "Launch" in e.name or "Memcpy" in e.name or "Synchronize" in e.name])
except Exception as e:
    print("ERR", repr(e))
