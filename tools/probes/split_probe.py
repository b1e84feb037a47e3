"""Probe: one frame's rays split into P parts on P HIP streams (P workspaces) vs one stream."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import ctypes as C
import torch
from arah_release_amd import config, hip, synthetic

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
scene = synthetic.SyntheticScene(0)
model, cfg = config.build_synthetic_model("zju377_mono", device=dev)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
streams = [torch.cuda.Stream(dev) for _ in range(P)]
wss = [hip.Workspace(dev) for _ in range(P)]
orig = hip.render
lib = hip.load_library()


def render_split(frame, ws, sampling, cam_loc, dirs, near_far, pose34):
    cam, d, nf = hip._f32(cam_loc), hip._f32(dirs), hip._f32(near_far)
    n, S = d.shape[0], sampling.n_steps
    rgb = torch.empty(n, 3, device=dev); pcam = torch.empty(n, 3, device=dev)
    vol = torch.empty(n, dtype=torch.uint8, device=dev); acc = torch.empty(n, device=dev)
    dists = torch.empty(n, device=dev); conv = torch.empty(n, dtype=torch.uint8, device=dev)
    d_pose = hip._f32(pose34).reshape(-1)[:12].contiguous()
    per = ((n + P - 1) // P + 63) // 64 * 64
    main = torch.cuda.current_stream(dev)
    ev = torch.cuda.Event(); ev.record(main)
    for k in range(P):
        r0, r1 = k * per, min(n, (k + 1) * per)
        if r1 <= r0:
            continue
        buf = wss[k].ensure(per, S)
        st = streams[k]
        st.wait_event(ev)
        m = r1 - r0
        rc = lib.arah_render(C.byref(frame.handle), C.byref(sampling.handle), hip._ptr(cam), C.c_int32(n),
                             hip._ptr(d[r0:r1]), hip._ptr(nf[r0:r1]), hip._ptr(d_pose), C.c_int32(m), hip._ptr(rgb[r0:r1]),
                             hip._ptr(pcam[r0:r1]), hip._ptr(vol[r0:r1]), hip._ptr(acc[r0:r1]), hip._ptr(dists[r0:r1]),
                             hip._ptr(conv[r0:r1]), hip._ptr(buf), C.c_size_t(buf.numel()), C.c_void_p(st.cuda_stream))
        assert rc == 0, rc
    for k in range(P):
        e = torch.cuda.Event(); e.record(streams[k]); main.wait_event(e)
    return rgb, pcam, vol, acc, dists, conv


inputs = [scene.make_inputs(512, 512, frame_idx=f, device=dev) for f in range(6)]


def run(tag):
    outs = []
    with torch.no_grad():
        for i in inputs[:2]:
            model(i, eval=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in inputs[2:]:
            outs.append(model(i, eval=True)["rgb_values"])
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    print(tag, "ms/frame %.2f" % (1e3 * dt))
    return outs


a = run("one stream")
hip.render = render_split
import arah_release_amd.renderer as R
R.hip.render = render_split
b = run("%d streams" % P)
print("bit-identical:", all(torch.equal(x, y) for x, y in zip(a, b)))
