"""Where the test.py frame's mesh branch spends its time (synchronised segments), and a first-rows diff of the marching-cubes
kernel against the tensor formulation."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arah_release_amd import config, hip, meshing, synthetic, training
dev = torch.device("cuda:0")
N = 24
ax = torch.linspace(-1, 1, N)
X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
sdf = (torch.sqrt((X - 0.1) ** 2 + (Y + 0.05) ** 2 + (Z - 0.2) ** 2) - 0.55).to(dev)
ref = meshing.marching_cubes(sdf.cpu())
tris, n = hip.marching_cubes(sdf, 0.0, cap=ref.shape[0] + 10)
got = tris.cpu()
print("F ref", ref.shape[0], "n_dev", int(n), "zero rows in got[:F]", int((got[:ref.shape[0]] == 0).all(-1).all(-1).sum()))
print("ref[0:2]", ref[0:2].tolist())
print("got[0:2]", got[0:2].tolist())
d = (got[:ref.shape[0]] - ref).abs()
print("max abs diff", float(d.max()), "rows equal", int((got[:ref.shape[0]] == ref).all(-1).all(-1).sum()))
gr = meshing.marching_cubes(sdf)   # tensor formulation on the GPU
print("tensor formulation on the GPU vs CPU: rows equal", int((gr.cpu() == ref).all(-1).all(-1).sum()), "of", ref.shape[0])

model, cfg = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
scene = synthetic.SyntheticScene(0)
inputs = scene.make_inputs(512, 512, frame_idx=1, device=dev)
with torch.no_grad():
    for _ in range(2):
        model(inputs, gen_cano_mesh=True, eval=True)
    frame, ws = model.idhr_network.last_frame, model.idhr_network.ray_tracer.workspace(dev)

    def seg(name, fn, reps=3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        print("%-28s %7.2f ms" % (name, 1e3 * (time.perf_counter() - t0) / reps))
        return out
    seg("render only", lambda: model(inputs, eval=True))
    seg("render + mesh branch", lambda: model(inputs, gen_cano_mesh=True, eval=True))
    g = seg("arah_sdf_grid 256^3", lambda: hip.sdf_grid(frame, ws, 256))
    seg("marching cubes (kernel)", lambda: hip.marching_cubes(g, 0.0, 1 << 20))
    seg("marching cubes (tensor ops)", lambda: meshing.marching_cubes(g))
    tri, nd = hip.marching_cubes(g, 0.0, 1 << 20)
    seg("whole branch, given nothing", lambda: meshing.canonical_mesh_outputs(frame, ws, inputs, want_tri=False))
    x_hat = training.unnormalize_canonical_points(tri.reshape(1, -1, 3), inputs["coord_min"][:1], inputs["coord_max"][:1], inputs["center"][:1])[0]
    seg("skin_lbs_counted", lambda: hip.skin_lbs_counted(frame, ws, x_hat, nd, per_item=3))
    xb = hip.skin_lbs_counted(frame, ws, x_hat, nd, per_item=3)
    posed = (xb + inputs["trans"].reshape(1, 3)).reshape(-1, 3, 3)
    cam_rot, cam_trans, K = inputs["cam_rot"][0], inputs["cam_trans"][0], inputs["intrinsics"][0]
    uvz = seg("project_opencv", lambda: meshing.project_opencv(posed, cam_rot, cam_trans, K))
    p2f = seg("rasterize posed", lambda: hip.rasterize(uvz, 512, 512))
    nrm = seg("face normals", lambda: meshing.face_normals(posed))
    seg("normal image", lambda: meshing.normal_image(p2f, nrm, -1.0))
    uv2 = seg("project_lookat", lambda: meshing.project_lookat(tri, 0.0, 512))
    seg("rasterize canonical", lambda: hip.rasterize(uv2, 512, 512, z_near=1.0))
    print("triangles", int(nd))
