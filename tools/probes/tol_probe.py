import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import golden, get_model
from arah_release_amd import hip, renderer, synthetic
scene = synthetic.SyntheticScene(0)
dev = torch.device("cuda:0")
model, cfg = get_model("zju377_mono", dev)
inputs = scene.make_inputs(64, 64, frame_idx=0, device=dev)
g = golden("f3_sdf.npz"); g2 = golden("f2_pointwise.npz")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for eng in ("split", "fp32"):
    os.environ["ARAH_PRECISION"] = eng
    with torch.no_grad():
        dec = model.sdf_decoder({"coords": torch.zeros(1, 1, 3, device=dev), "rots": inputs["rots"][:1], "Jtrs": inputs["Jtrs"][:1], "latent": model.latent(inputs["geo_latent_code_idx"])})
        pc = dict(inputs["pose_cond"]); pc["latent_code"] = model.latent(pc["latent_code_idx"])
        fr = renderer.build_frame(dec["decoder"], model.skinning_model, model.color_decoder, model.deviation_decoder, pc, inputs["smpl_verts"], inputs["skinning_weights"], inputs["bone_transforms"], inputs["trans"], inputs["coord_min"], inputs["coord_max"], inputs["center"])
    ws = hip.Workspace(dev)
    sdf, feat, grad = hip.sdf_eval(fr, ws, T(g["x_norm"]), want_feat=True, want_grad=True)
    def rep(name, a, b):
        a = a.cpu().numpy(); d = np.abs(a - b)
        print(eng, name, "max abs %.3e" % d.max(), "max |ref| %.3e" % np.abs(b).max(), "worst excess over rtol1e-4: atol needed %.3e" % (d - 1e-4 * np.abs(b)).max())
    rep("sdf", sdf, g["sdf"]); rep("feat", feat, g["feat"]); rep("grad", grad, g["grad"])
    w, xb, Tm = hip.skin_lbs(fr, ws, T(g2["x_hat"]))
    rep("w", w, g2["weights"]); rep("xbar", xb, g2["x_bar"]); rep("T", Tm, g2["T"])
    jac = hip.skin_jacobian(fr, ws, T(g2["x_hat"])); rep("jac", jac, g2["jac"])
