"""Throughput of the unit seams (one kernel each) on the current GPU: points/s and fp32-MFMA fraction."""
import sys, os, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as g
g.build()
from arah_release_amd import config, hip, renderer, synthetic

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "zju377_mono"
P = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2_000_000
model, cfg = config.build_synthetic_model(name, device=dev)
scene = synthetic.SyntheticScene(0)
inputs = scene.make_inputs(64, 64, frame_idx=0, device=dev)
with torch.no_grad():
    dec = model.sdf_decoder({"coords": torch.zeros(1, 1, 3, device=dev), "rots": inputs["rots"][:1],
                             "Jtrs": inputs["Jtrs"][:1], "latent": model.latent(inputs["geo_latent_code_idx"])})
    pose_cond = dict(inputs["pose_cond"]); pose_cond["latent_code"] = model.latent(pose_cond["latent_code_idx"])
    frame = renderer.build_frame(dec["decoder"], model.skinning_model, model.color_decoder, model.deviation_decoder,
                                 pose_cond, inputs["smpl_verts"], inputs["skinning_weights"], inputs["bone_transforms"],
                                 inputs["trans"], inputs["coord_min"], inputs["coord_max"], inputs["center"])
ws = hip.Workspace(dev)
gen = torch.Generator(device=dev).manual_seed(0)
xn = torch.rand(P, 3, device=dev, generator=gen) * 1.6 - 0.8
cmin, cmax = float(inputs["coord_min"].reshape(-1)[0]), float(inputs["coord_max"].reshape(-1)[0])
xh = (xn / 2 + 0.5) * 1.1 * (cmax - cmin) + cmin - (cmax - cmin) * 0.05 + inputs["center"][0, 0]
verts = inputs["smpl_verts"][0]
pts = verts[torch.randint(0, verts.shape[0], (P,), device=dev, generator=gen)] + torch.randn(P, 3, device=dev, generator=gen) * 0.03
nrm = torch.randn(P, 3, device=dev, generator=gen)
feat = torch.rand(P, 256, device=dev, generator=gen) * 2 - 1


def timeit(fn, flops_per_pt, label, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    tf = P * flops_per_pt / dt / 1e12
    print("%-28s %8.2f ms  %8.1f Mpts/s  %7.1f TFLOP/s  (%.0f%% of 157.3)" % (label, dt * 1e3, P / dt / 1e6, tf, 100 * tf / 157.3))


mode = cfg["model"]["renderer_kwargs"]["mode"]
fcol = 794112 if mode == "no_view_dir" else 821760
timeit(lambda: hip.sdf_eval(frame, ws, xn), 657408, "sdf_eval fwd")
timeit(lambda: hip.sdf_eval(frame, ws, xn, want_grad=True), 2 * 657408, "sdf_eval fwd+grad")
timeit(lambda: hip.skin_lbs(frame, ws, xh), 105472, "skin_lbs")
timeit(lambda: hip.skin_jacobian(frame, ws, xh[: P // 4]), 105472, "skin_jacobian (P/4, x4 cols)")
timeit(lambda: hip.color_eval(frame, ws, xn, nrm, nrm, feat), fcol, "color_eval")
timeit(lambda: hip.nearest_inverse_lbs(frame, ws, pts), 55120, "nearest_inverse_lbs")
tgt = hip.skin_lbs(frame, ws, xh)[1]
x0 = xh + torch.randn(P, 3, device=dev, generator=gen) * 0.01
T0 = torch.eye(4, device=dev).expand(P, 4, 4).contiguous()
ws.reset_counters()
timeit(lambda: hip.broyden3_lbs(frame, ws, tgt, x0, T0), 105472, "broyden3 (flops/pt nominal x1)", reps=1)
c = ws.counters()
print("broyden3 skin evals per point: %.2f" % (c["n_skin_fwd"] / (2 * P)))
