"""Timing ablations of the 256-wide forward SDF trunk (split engine) through the arah_sdf_eval seam:
    ARAH_LIB_PATH=tools/ubench/bin/libarah_abl_X.so python tools/ablate_trunk.py [n_points]
The ablated builds (-DARAH_ABL_*, csrc/mlp.hpp) compute WRONG values; only the launch time is of interest.  Points are
uniform in the normalised cube, the frame is the synthetic subject's."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from arah_release_amd import config, hip, renderer, synthetic  # noqa: E402

density = "--density" in sys.argv          # time k_density (the product's pass, ARAH_DENSITY_TILE = 128 | 64) instead of k_sdf_eval
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 7_300_000
dev = torch.device("cuda:0")
model, cfg = config.build_synthetic_model("zju377_mono", device=dev)
scene = synthetic.SyntheticScene(0)
inputs = scene.make_inputs(64, 64, frame_idx=0, device=dev)
with torch.no_grad():
    dec = model.sdf_decoder({"coords": torch.zeros(1, 1, 3, device=dev), "rots": inputs["rots"][:1], "Jtrs": inputs["Jtrs"][:1],
                             "latent": model.latent(inputs["geo_latent_code_idx"])})
    pose_cond = dict(inputs["pose_cond"])
    pose_cond["latent_code"] = model.latent(pose_cond["latent_code_idx"])
    frame = renderer.build_frame(dec["decoder"], model.skinning_model, model.color_decoder, model.deviation_decoder, pose_cond,
                                 inputs["smpl_verts"], inputs["skinning_weights"], inputs["bone_transforms"], inputs["trans"],
                                 inputs["coord_min"], inputs["coord_max"], inputs["center"])
ws = hip.Workspace(dev)
x = torch.rand(n, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) * 2 - 1
if density:
    S = 64
    nr = n // S
    samp = hip.Sampling(dev, S, 16, 16)
    dirs = torch.zeros(nr, 3, device=dev)
    dirs[:, 2] = 1.0
    z = torch.linspace(1.0, 2.0, S, device=dev).repeat(nr, 1).contiguous()
    pts = x[:nr * S].reshape(nr, S, 3).contiguous()
    Tm = torch.eye(4, device=dev).repeat(nr, S, 1, 1).contiguous()
    mask = torch.ones(nr, S, dtype=torch.uint8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for it in range(6):
        samp.set_events("density", e0, e1)
        hip.shade_composite(frame, ws, samp, dirs, z, pts, Tm, mask)
        torch.cuda.synchronize()
        samp.set_events("density", None, None)
        if it:
            ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    print("%-28s k_density tile %-3s %8.3f ms  %7.1f algorithmic TFLOP/s" % (
        os.path.basename(os.environ.get("ARAH_LIB_PATH", "libarah_hip.so")), os.environ.get("ARAH_DENSITY_TILE", "128"), ms,
        nr * S * 657408 / ms / 1e9))
    sys.exit(0)
hip.sdf_eval(frame, ws, x)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.sdf_eval(frame, ws, x)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[len(ts) // 2]
print("%-28s %8.3f ms  %7.1f algorithmic TFLOP/s" % (os.path.basename(os.environ.get("ARAH_LIB_PATH", "libarah_hip.so")), ms,
                                                    n * 657408 / ms / 1e9))
