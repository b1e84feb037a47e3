import sys, os, torch
sys.path.insert(0, os.getcwd())
from arah_release_amd import config, hip, renderer, synthetic
dev = torch.device("cuda:0")
scene = synthetic.SyntheticScene(0)
model, cfg = config.build_synthetic_model("zju377_mono", device=dev)
inputs = scene.make_inputs(512, 512, frame_idx=7, device=dev)
with torch.no_grad():
    dec = model.sdf_decoder({"coords": torch.zeros(1, 1, 3, device=dev), "rots": inputs["rots"][:1],
                             "Jtrs": inputs["Jtrs"][:1], "latent": model.latent(inputs["geo_latent_code_idx"])})
    pose_cond = dict(inputs["pose_cond"]); pose_cond["latent_code"] = model.latent(pose_cond["latent_code_idx"])
    frame = renderer.build_frame(dec["decoder"], model.skinning_model, model.color_decoder, model.deviation_decoder, pose_cond,
                                 inputs["smpl_verts"], inputs["skinning_weights"], inputs["bone_transforms"], inputs["trans"],
                                 inputs["coord_min"], inputs["coord_max"], inputs["center"])
ws = hip.Workspace(dev)
samp = hip.Sampling(dev, 64, 16, 16, cfg["model"]["cano_view_dirs"], False)
pose = torch.eye(4)[:3]
cam, d, nf = inputs["cam_loc"], inputs["ray_dirs"][0], inputs["body_bounds_intersections"][0]
N = d.shape[0]
full = hip.render(frame, ws, samp, cam, d, nf, pose)
half = N // 2
a = hip.render(frame, ws, samp, cam, d[:half].contiguous(), nf[:half].contiguous(), pose)
b = hip.render(frame, ws, samp, cam, d[half:].contiguous(), nf[half:].contiguous(), pose)
names = ["rgb", "pcam", "vol", "acc", "dists", "conv"]
for i, nm in enumerate(names):
    cat = torch.cat([a[i], b[i]])
    neq = (cat != full[i])
    if neq.ndim > 1: neq = neq.any(-1)
    md = (cat.float() - full[i].float()).abs().max().item()
    print(nm, "mismatching rays:", int(neq.sum()), "max diff", md, "first idx", neq.nonzero()[:5].flatten().tolist())
# tracer only
t_full = hip.trace(frame, ws, cam, d, nf)
t_a = hip.trace(frame, ws, cam, d[:half].contiguous(), nf[:half].contiguous())
for i in range(len(t_full)):
    x, y = t_full[i][:half], t_a[i]
    neq = (x != y)
    if neq.ndim > 1: neq = neq.flatten(1).any(-1)
    print("trace out", i, "mismatch", int(neq.sum()))
full2 = hip.render(frame, ws, samp, cam, d, nf, pose)
for i, nm in enumerate(names):
    print("rerun", nm, "mismatch", int((full2[i] != full[i]).sum()))
t2 = hip.trace(frame, ws, cam, d, nf)
for i in range(len(t2)):
    print("retrace", i, int((t2[i] != t_full[i]).sum()))
# small subsets of different sizes: first 4096 rays alone vs within first 20000
q = 4096
t_q = hip.trace(frame, ws, cam, d[:q].contiguous(), nf[:q].contiguous())
t_r = hip.trace(frame, ws, cam, d[:20000].contiguous(), nf[:20000].contiguous())
for i in range(len(t_q)):
    print("narrow-vs-wide trace", i, int((t_q[i] != t_r[i][:q]).sum()), "vs full", int((t_q[i] != t_full[i][:q]).sum()))
