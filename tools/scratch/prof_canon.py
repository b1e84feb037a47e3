"""Phase clocks of loop C (needs a library built with -DARAH_PROFILE_CANON)."""
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
hip.render(frame, ws, samp, cam, d, nf, pose)
ws.reset_counters()
hip.render(frame, ws, samp, cam, d, nf, pose)
torch.cuda.synchronize()
c = ws.counters()
print(c)
if "reserved" in c:
    tiles = c["n_skin_fwd"] / 64.0
    print("loop C, wave 0, per 64-point tile: skin MLP %.0f clocks, per-point tail %.0f clocks (tiles %.0f)" %
          (c["reserved"][0] / tiles, c["reserved"][1] / tiles, tiles))
