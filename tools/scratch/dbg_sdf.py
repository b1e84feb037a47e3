import sys, os, torch
sys.path.insert(0, os.getcwd())
from arah_release_amd import config, hip, renderer, synthetic
dev = torch.device("cuda:0")
scene = synthetic.SyntheticScene(0)
model, cfg = config.build_synthetic_model("zju377_mono", device=dev)
inputs = scene.make_inputs(64, 64, frame_idx=7, device=dev)
with torch.no_grad():
    dec = model.sdf_decoder({"coords": torch.zeros(1, 1, 3, device=dev), "rots": inputs["rots"][:1],
                             "Jtrs": inputs["Jtrs"][:1], "latent": model.latent(inputs["geo_latent_code_idx"])})
    pose_cond = dict(inputs["pose_cond"]); pose_cond["latent_code"] = model.latent(pose_cond["latent_code_idx"])
    frame = renderer.build_frame(dec["decoder"], model.skinning_model, model.color_decoder, model.deviation_decoder, pose_cond,
                                 inputs["smpl_verts"], inputs["skinning_weights"], inputs["bone_transforms"], inputs["trans"],
                                 inputs["coord_min"], inputs["coord_max"], inputs["center"])
ws = hip.Workspace(dev)
g = torch.Generator().manual_seed(3)
x = (torch.rand(200000, 3, generator=g) * 2 - 1).to(dev)
for grad in (False, True):
    a = hip.sdf_eval(frame, ws, x, want_feat=True, want_grad=grad)
    b = hip.sdf_eval(frame, ws, x, want_feat=True, want_grad=grad)
    print("grad", grad, "rerun sdf mismatch", int((a[0] != b[0]).sum()), "feat", int((a[1] != b[1]).sum()),
          "grad", 0 if not grad else int((a[2] != b[2]).sum()))
    perm = torch.randperm(x.shape[0], generator=g).to(dev)
    c = hip.sdf_eval(frame, ws, x[perm].contiguous(), want_feat=True, want_grad=grad)
    print("   perm sdf mismatch", int((c[0] != a[0][perm]).sum()), "feat", int((c[1] != a[1][perm]).sum()),
          "maxdiff", float((c[0] - a[0][perm]).abs().max()))
    s = hip.sdf_eval(frame, ws, x[:1000].contiguous(), want_feat=True, want_grad=grad)
    print("   subset sdf mismatch", int((s[0] != a[0][:1000]).sum()))
a = hip.sdf_eval(frame, ws, x)
b = hip.sdf_eval(frame, ws, x)
print("no-feat rerun sdf mismatch", int((a[0] != b[0]).sum()), "max", float((a[0]-b[0]).abs().max()))
