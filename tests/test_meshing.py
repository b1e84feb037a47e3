"""gen_cano_mesh branch (reference models/__init__.py:203-311, utils/sdf_meshing.py:13-114).

CPU: the marching-cubes case table and the tensorised extraction (they are plain torch and run on any device);
GPU: the lattice SDF launch, the rasteriser against the numpy restatement in oracle/mesh_oracle.py, and the whole
branch through MetaAvatarRender.forward(gen_cano_mesh=True)."""
import math
from collections import Counter

import numpy as np
import pytest
import torch

from conftest import get_model, golden

gpu = pytest.mark.gpu


def test_case_table_uses_exactly_the_crossed_edges():
    from arah_release_amd import meshing
    table, ntri = meshing.case_table()
    assert table.shape == (256, 15) and ntri.max() == 5 and ntri[0] == 0 and ntri[255] == 0
    for case in range(256):
        inside = [(case >> c) & 1 for c in range(8)]
        crossed = {i for i, (a, b) in enumerate(meshing.EDGES) if inside[a] != inside[b]}
        used = set(int(e) for e in table[case] if e >= 0)
        assert used == crossed, case


def _edge_multiplicities(tri):
    q = torch.round(tri.double() * 1e6).long().tolist()
    cnt = Counter()
    for f in q:
        for i in range(3):
            cnt[frozenset((tuple(f[i]), tuple(f[(i + 1) % 3])))] += 1
    return Counter(cnt.values())


def test_marching_cubes_sphere_is_watertight_and_oriented():
    from arah_release_amd import meshing
    N = 40
    ax = torch.linspace(-1, 1, N)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    c = torch.tensor([0.1, -0.05, 0.2])
    sdf = torch.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) - 0.55
    tri = meshing.marching_cubes(sdf)
    assert tri.shape[0] > 1000
    assert set(_edge_multiplicities(tri)) == {2}                       # closed 2-manifold: every edge twice
    assert float(((tri - c).norm(dim=-1) - 0.55).abs().max()) < 2e-3   # vertices on the level set (linear interpolation)
    nrm = meshing.face_normals(tri)
    out = tri.mean(1) - c
    assert bool(((nrm * out).sum(1) < 0).all())                        # skimage 'descent': towards decreasing values
    # the lattice indexing of sdf_meshing.py: an off-centre sphere must come out where it was put
    assert float((tri.reshape(-1, 3).mean(0) - c).abs().max()) < 0.02


def test_marching_cubes_two_touching_blobs_has_no_cracks():
    """Ambiguous faces (four crossings) are cut the same way from both sides."""
    from arah_release_amd import meshing
    N = 24
    ax = torch.linspace(-1, 1, N)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    g = torch.Generator().manual_seed(3)
    sdf = torch.sqrt(X ** 2 + Y ** 2 + Z ** 2) - 0.6 + 0.35 * torch.randn(N, N, N, generator=g)
    sdf[0], sdf[-1], sdf[:, 0], sdf[:, -1], sdf[:, :, 0], sdf[:, :, -1] = 1, 1, 1, 1, 1, 1   # closed inside the volume
    mult = _edge_multiplicities(meshing.marching_cubes(sdf))
    assert 1 not in mult and 3 not in mult                              # no boundary edges: no cracks


@gpu
@pytest.mark.parametrize("case", ["sphere", "noise", "emitted_sdf"])
def test_marching_cubes_kernel_is_the_tensor_formulation(scene, case):
    """arah_marching_cubes (csrc/mcubes.hpp) against meshing.marching_cubes, the tensor formulation it replaces, run on the
    CPU: the same number of triangles, in the same order, every corner bit-equal; the orientation may differ where a
    triangle is degenerate (its normal is rounding noise), nowhere else.  Rows beyond the count are zero, the count stays
    on the device, a capacity below the count truncates and reports the full count."""
    from arah_release_amd import hip, meshing
    dev = torch.device("cuda:0")
    if case == "emitted_sdf":
        model, cfg = get_model("zju377_mono", dev)
        inputs = scene.make_inputs(512, 512, frame_idx=1, device=dev, max_rays=1024)
        with torch.no_grad():
            model(inputs, eval=True)
        frame, ws = model.idhr_network.last_frame, model.idhr_network.ray_tracer.workspace(dev)
        sdf = hip.sdf_grid(frame, ws, 256)
    else:
        N = 56 if case == "sphere" else 33
        ax = torch.linspace(-1, 1, N)
        X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
        sdf = torch.sqrt((X - 0.1) ** 2 + (Y + 0.05) ** 2 + (Z - 0.2) ** 2) - 0.55
        if case == "noise":
            sdf = sdf + 0.4 * torch.randn(N, N, N, generator=torch.Generator().manual_seed(5))
        sdf = sdf.to(dev)
    ref = meshing.marching_cubes(sdf.cpu())
    F = ref.shape[0]
    assert F > 1000
    tris, n_dev = hip.marching_cubes(sdf, 0.0, cap=F + 1000)
    assert int(n_dev.item()) == F
    got = tris.cpu()
    assert bool((got[F:] == 0).all())
    got = got[:F]
    same = (got == ref).all(-1).all(-1)
    flipped = (got[:, [0, 2, 1]] == ref).all(-1).all(-1)
    assert bool((same | flipped).all())                                 # corners bit-equal, triangle for triangle
    area = torch.cross(ref[:, 1] - ref[:, 0], ref[:, 2] - ref[:, 0], dim=1).norm(dim=1)
    assert bool((area[~same] < 1e-9).all()), int((~same).sum())          # orientation differs on degenerate triangles only
    small, n2 = hip.marching_cubes(sdf, 0.0, cap=F // 2)
    assert int(n2.item()) == F and torch.equal(small.cpu()[:F // 2][same[:F // 2]], ref[:F // 2][same[:F // 2]])


@gpu
def test_mesh_branch_keeps_its_triangle_count_on_the_device(scene):
    """The model entry's gen_cano_mesh branch must not wait for the GPU (frames of a test sequence overlap): the triangle
    count travels to the host asynchronously, and the maps of the fixed-capacity path equal those of the trimmed mesh."""
    from arah_release_amd import meshing
    dev = torch.device("cuda:0")
    model, cfg = get_model("zju377_mono", dev)
    inputs = scene.make_inputs(512, 512, frame_idx=2, device=dev, max_rays=2048)
    with torch.no_grad():
        out = model(inputs, gen_cano_mesh=True, eval=True)
    n, overflowed = meshing.mesh_counts(dev)
    assert n is not None and 20000 < n < meshing.MC_DEFAULT_CAP and overflowed == 0
    frame, ws = model.idhr_network.last_frame, model.idhr_network.ray_tracer.workspace(dev)
    maps, tri = meshing.canonical_mesh_outputs(frame, ws, inputs)
    assert tri.shape[0] == n
    maps2, _ = meshing.canonical_mesh_outputs(frame, ws, inputs, tri=tri)   # exact-size mesh through arah_skin_lbs
    for k in ("output_normal", "normal_cano_front", "normal_cano_back"):
        assert torch.equal(maps[k], out[k]) and torch.equal(maps[k], maps2[k]), k


def test_lookat_projection_matches_the_documented_camera():
    """look_at_view_transform(2, 0, azim) + FoVPerspectiveCameras(fov 60): +x is right / +y is up seen from the front,
    mirrored in x seen from the back; the optical axis hits the image centre."""
    from arah_release_amd import meshing
    f = 1.0 / math.tan(math.radians(30.0))
    p = torch.tensor([[0.0, 0.0, 0.0], [0.3, 0.2, 0.1]])
    front = meshing.project_lookat(p, 0.0, 512)
    back = meshing.project_lookat(p, 180.0, 512)
    np.testing.assert_allclose(front[0].numpy(), [256.0, 256.0, 2.0], atol=1e-4)
    np.testing.assert_allclose(front[1].numpy(), [(1 + f * 0.3 / 1.9) * 256, (1 - f * 0.2 / 1.9) * 256, 1.9], rtol=1e-5)
    np.testing.assert_allclose(back[1].numpy(), [(1 - f * 0.3 / 2.1) * 256, (1 - f * 0.2 / 2.1) * 256, 2.1], rtol=1e-5)


def test_rasterizer_oracle_on_two_overlapping_triangles():
    from oracle.mesh_oracle import rasterize_np
    tri = np.array([[[1, 1, 2.0], [7, 1, 2.0], [1, 7, 2.0]], [[0, 0, 1.0], [5, 0, 1.0], [0, 5, 3.0]]], np.float32)
    p2f = rasterize_np(tri, 8, 8)
    assert p2f[1, 1] == 1 and p2f[5, 1] == 0 and p2f[7, 7] == -1      # nearer face wins where both cover
    assert p2f[3, 1] in (0, 1)


# ---------------------------------------------------------------------------------------------------------- GPU
@gpu
def test_rasterize_against_oracle():
    from arah_release_amd import hip
    from oracle.mesh_oracle import rasterize_np
    g = torch.Generator().manual_seed(5)
    F, H, W = 600, 96, 128
    c = torch.rand(F, 1, 2, generator=g) * torch.tensor([W, H]) * 1.2 - torch.tensor([W, H]) * 0.1
    uv = c + (torch.rand(F, 3, 2, generator=g) - 0.5) * 14
    z = 1.0 + torch.rand(F, 3, 1, generator=g) * 3
    z[:7] = -0.5                                                          # behind the camera: dropped
    tri = torch.cat([uv, z], dim=-1).float()
    got = hip.rasterize(tri.cuda(), H, W).cpu().numpy()
    ref = rasterize_np(tri.numpy(), H, W)
    assert got.shape == (H, W) and (got >= 0).mean() > 0.3
    assert (got == ref).mean() >= 0.999                                  # fp32 contraction at triangle edges only
    assert not np.isin(got, np.arange(7)).any()


def _f15():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f15_sdf_lattice.npz"))


def test_lattice_convention_is_the_references():
    """Fixture F15 (the reference's own create_mesh_vertices_and_faces with a recording decoder and a recording stand-in
    for skimage): the volume handed to marching cubes is indexed [ix, iy, iz] with ix slowest, on coordinates
    index * (2 / (N - 1)) - 1 formed by a float32 product and a float32 sum; marching cubes runs at level 0 with that
    spacing and its vertices (index * spacing) map back through origin + vertex.  The build's marching_cubes restates the
    last step; its lattice restates the first (checked on the device in test_sdf_grid_is_the_references_lattice)."""
    from arah_release_amd import meshing
    g = _f15()
    N = int(g["N"])
    vs = np.float32(2.0 / (N - 1))
    ax = (np.arange(N, dtype=np.float32) * vs + np.float32(-1.0)).astype(np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    np.testing.assert_array_equal(np.stack([X, Y, Z], -1).reshape(-1, 3), g["coords"])            # order and bits
    f = lambda c: (0.3 * c[..., 0] - 0.5 * c[..., 1] + 0.7 * c[..., 2] + 0.11)
    np.testing.assert_allclose(g["volume"], f(g["coords"].reshape(N, N, N, 3)), rtol=0, atol=1e-6)     # volume[ix, iy, iz]
    assert float(g["level"]) == 0.0 and np.allclose(g["spacing"], 2.0 / (N - 1))
    np.testing.assert_allclose(g["mesh_points"], g["verts_idx"] * (2.0 / (N - 1)) - 1.0, atol=1e-12)
    np.testing.assert_allclose(g["mesh_points_scaled"], (g["verts_idx"] * (2.0 / (N - 1)) - 1.0) / 2.0 - np.array([0.1, -0.2, 0.3]),
                               atol=1e-12)
    ax256 = (np.arange(256, dtype=np.float32) * np.float32(2.0 / 255) + np.float32(-1.0)).astype(np.float32)
    for k in ("axis256", "axis256_y", "axis256_z"):
        np.testing.assert_array_equal(g[k], ax256)
    assert g["batches256"].tolist() == [64 ** 3] * 64
    # the build's extraction on the reference's own volume: the plane 0.3 x - 0.5 y + 0.7 z + 0.11 = 0, vertices in the
    # reference's coordinates (origin + index * spacing)
    tri = meshing.marching_cubes(torch.from_numpy(g["volume"]))
    assert tri.shape[0] > 0 and float(tri.abs().max()) <= 1.0 + 1e-6
    v = tri.reshape(-1, 3).double().numpy()
    assert np.abs(f(v)).max() < 1e-5                                        # on the level set, in the reference's frame
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).astype(np.float64)
    assert (n @ np.array([0.3, -0.5, 0.7]) < 0).all()                       # right-hand normals point down the gradient


@gpu
def test_sdf_grid_is_the_references_lattice(scene):
    """arah_sdf_grid against fixture F15: the value at [ix, iy, iz] is the SDF at the coordinates the reference asks its
    decoder for at that position of its volume -- bit for bit the same inputs, hence the same outputs as a point query."""
    from arah_release_amd import hip
    g = _f15()
    dev = torch.device("cuda:0")
    model, cfg = get_model("zju377_mono", dev)
    with torch.no_grad():
        model(scene.make_inputs(32, 32, frame_idx=1, device=dev), eval=True)
    frame, ws = model.idhr_network.last_frame, model.idhr_network.ray_tracer.workspace(dev)
    N = int(g["N"])
    grid = hip.sdf_grid(frame, ws, N)
    sdf, _, _ = hip.sdf_eval(frame, ws, torch.from_numpy(g["coords"]).to(dev))
    assert torch.equal(grid.reshape(-1), sdf)
    big = hip.sdf_grid(frame, ws, 256)
    rng = np.random.RandomState(0)
    idx = rng.randint(0, 256, size=(4096, 3))
    ax = torch.from_numpy(g["axis256"]).to(dev)
    pts = torch.stack([ax[idx[:, 0]], ax[idx[:, 1]], ax[idx[:, 2]]], dim=-1)
    sdf, _, _ = hip.sdf_eval(frame, ws, pts)
    assert torch.equal(big[idx[:, 0], idx[:, 1], idx[:, 2]], sdf)


@gpu
def test_sdf_grid_matches_point_queries(scene):
    from arah_release_amd import hip
    dev = torch.device("cuda:0")
    model, cfg = get_model("zju377_mono", dev)
    inputs = scene.make_inputs(32, 32, frame_idx=1, device=dev)
    with torch.no_grad():
        out = model(inputs, eval=True)
    frame, ws = model.idhr_network.last_frame, model.idhr_network.ray_tracer.workspace(dev)
    N = 20
    grid = hip.sdf_grid(frame, ws, N)
    ax = torch.arange(N, device=dev, dtype=torch.float32) * (2.0 / (N - 1)) + -1.0
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    pts = torch.stack([X, Y, Z], dim=-1).reshape(-1, 3)
    sdf, _, _ = hip.sdf_eval(frame, ws, pts)
    # same lattice, coordinates formed on the host here: agreement to the last bits of the SIREN's input sensitivity
    assert float((grid.reshape(-1) - sdf).abs().max()) < 2e-6


@gpu
@pytest.mark.parametrize("name", ["zju377_mono", "h36m"])
def test_gen_cano_mesh_branch(scene, name):
    """MetaAvatarRender.forward(inputs, gen_cano_mesh=True, eval=True) as lightning_model.py:320 calls it: the three
    normal maps, composed from the build's own pieces checked one by one (mesh on the SDF's zero set, rasteriser vs
    the oracle) -- plus what can be said without pytorch3d / skimage: silhouettes, orientation, unit normals."""
    from arah_release_amd import meshing
    dev = torch.device("cuda:0")
    model, cfg = get_model(name, dev)
    # the reference rasterises the posed mesh at 512 x 512 whatever the frame size (models/__init__.py:226-233): the
    # intrinsics must be those of a 512 x 512 frame; 2048 of its rays are enough for the rendering half of the call
    inputs = scene.make_inputs(512, 512, frame_idx=2, device=dev, max_rays=2048)
    with torch.no_grad():
        out = model(inputs, gen_cano_mesh=True, eval=True)
    for k in ("output_normal", "normal_cano_front", "normal_cano_back"):
        assert tuple(out[k].shape) == (1, 512, 512, 3) and out[k].dtype == torch.float32
        assert float(out[k].min()) >= 0.0 and float(out[k].max()) <= 1.0
    assert {"points_cam", "network_body_mask", "rgb_values", "sdf_params"} <= set(out.keys())
    front, back, posed = out["normal_cano_front"][0], out["normal_cano_back"][0], out["output_normal"][0]
    fg_f = (front - 0.5).abs().sum(-1) > 1e-6                            # background is (0.5, 0.5, 0.5)
    fg_b = (back - 0.5).abs().sum(-1) > 1e-6
    assert 0.03 < float(fg_f.float().mean()) < 0.6
    # front and back views see the same body mirrored left-right (cameras at +-2 on the z axis); perspective makes the
    # nearer parts a little larger in each view, so the silhouettes agree up to a rim
    assert float((fg_f == fg_b.flip(1)).float().mean()) > 0.93
    # un-negated face normals point INTO the body (skimage 'descent' orientation): away from the front camera
    # (z < 0) on the front map, towards +z on the back map
    nf = front[fg_f] * 2 - 1
    nb = back[fg_b] * 2 - 1
    assert float((nf[:, 2] < 0).float().mean()) > 0.99 and float((nb[:, 2] > 0).float().mean()) > 0.99
    assert float((nf.norm(dim=-1) - 1).abs().max()) < 1e-3
    # the mesh lies on the zero set of the emitted SDF and skins onto the posed body
    frame, ws = model.idhr_network.last_frame, model.idhr_network.ray_tracer.workspace(dev)
    from arah_release_amd import hip
    maps, tri = meshing.canonical_mesh_outputs(frame, ws, inputs)
    assert torch.equal(maps["normal_cano_front"], out["normal_cano_front"])
    sdf, _, _ = hip.sdf_eval(frame, ws, tri.reshape(-1, 3))
    assert float(sdf.abs().max()) < 2e-3 and tri.shape[0] > 20000
    # posed normal map: foreground where the SMPL body projects, normals face the camera (z < 0 in OpenCV camera space)
    fg_p = (posed.sum(-1) > 1e-6)
    npos = posed[fg_p] * 2 - 1
    assert float((npos[:, 2] < 0).float().mean()) > 0.98
    K = inputs["intrinsics"][0]
    v = inputs["smpl_verts"][0]
    u = (K[0, 0] * v[:, 0] / v[:, 2] + K[0, 2]).long().clamp(0, 511)
    w = (K[1, 1] * v[:, 1] / v[:, 2] + K[1, 2]).long().clamp(0, 511)
    assert float(fg_p[w, u].float().mean()) > 0.9                        # body vertices land on the rendered silhouette


# ------------------------------------------------------------------------------------------ fixture F18
def test_normal_maps_of_the_reference_branch_cpu():
    """F18 (tests/golden/make_golden.py f18): the reference's OWN gen_cano_mesh branch (models/__init__.py:203-311) run with
    recorders in place of pytorch3d.  On the CPU: from the mesh it rasterised, the posed vertices IT computed and the
    pix_to_face images it was handed, the build's colouring (which normals, which sign, which frame, background, the [0,1]
    map) reproduces the reference's three 512 x 512 normal maps, and the build's projections + the rasteriser oracle reproduce
    the pix_to_face images from the camera arguments the reference passed on."""
    from arah_release_amd import meshing
    from oracle import mesh_oracle
    g = golden("f18_cano_mesh_branch.npz")
    tri = torch.from_numpy(g["tri"])
    posed = torch.from_numpy(g["posed_verts"]).reshape(-1, 3, 3)
    cam_rot, cam_trans, K = torch.from_numpy(g["cam_rot"])[0], torch.from_numpy(g["cam_trans"])[0], torch.from_numpy(g["intrinsics"])[0]
    # what the reference hands to cameras_from_opencv_projection is the dataset's camera, for a 512 x 512 image
    np.testing.assert_array_equal(g["opencv_R"][0], g["cam_rot"][0])
    np.testing.assert_array_equal(g["opencv_t"][0], g["cam_trans"][0])
    np.testing.assert_array_equal(g["opencv_K"][0], g["intrinsics"][0])
    np.testing.assert_array_equal(g["opencv_image_size"], [[512.0, 512.0]])
    p2f = {k: torch.from_numpy(g[k].astype(np.int64)) for k in ("p2f_posed", "p2f_front", "p2f_back")}
    assert int((p2f["p2f_posed"] >= 0).sum()) > 3000 and int((p2f["p2f_front"] >= 0).sum()) > 3000
    img = meshing.normal_image(p2f["p2f_posed"], -meshing.face_normals(posed) @ cam_rot.t(), -1.0)
    np.testing.assert_allclose(img.numpy(), g["output_normal"], rtol=0, atol=2e-6)
    n_cano = meshing.face_normals(tri)
    np.testing.assert_allclose(meshing.normal_image(p2f["p2f_front"], n_cano, 0.0).numpy(), g["normal_cano_front"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(meshing.normal_image(p2f["p2f_back"], n_cano, 0.0).numpy(), g["normal_cano_back"], rtol=0, atol=2e-6)
    # REGRESSION check only, not a pin to the reference: the recorded pix_to_face came from these same two functions
    # (meshing.project_opencv + oracle rasterize_np stand in for pytorch3d, which is absent here and from the reference tree);
    # what F18 pins to the reference is above -- posed vertices, camera arguments, normal selection / sign / frame / colouring
    uvz = meshing.project_opencv(posed, cam_rot, cam_trans, K).numpy()
    sub = mesh_oracle.rasterize_np(uvz, 512, 512)
    assert (sub == g["p2f_posed"]).all()


@gpu
def test_canonical_mesh_outputs_against_the_reference_branch(scene):
    """F18 on the device: the build's forward skinning of the mesh vertices against the posed vertices the reference computed
    (its unnormalisation, forward_skinning and translation: element-wise), arah_rasterize on the build's projections against
    the recorded pix_to_face (equal where the depth order is not a tie of the skinning's last bits), and the whole
    canonical_mesh_outputs on the injected mesh against the reference's three normal maps."""
    from arah_release_amd import config, hip, meshing, renderer, training
    g = golden("f18_cano_mesh_branch.npz")
    dev = torch.device("cuda:0")
    model, cfg = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
    inputs = scene.make_inputs(int(g["H"]), int(g["W"]), frame_idx=int(g["frame_idx"]), max_rays=int(g["max_rays"]), device=dev)
    with torch.no_grad():
        dec = model.sdf_decoder({"coords": torch.zeros(1, 1, 3, device=dev), "rots": inputs["rots"][:1],
                                 "Jtrs": inputs["Jtrs"][:1], "latent": model.latent(inputs["geo_latent_code_idx"])})
        pose_cond = dict(inputs["pose_cond"])
        pose_cond["latent_code"] = model.latent(pose_cond["latent_code_idx"])
        frame = renderer.build_frame(dec["decoder"], model.skinning_model, model.color_decoder, model.deviation_decoder,
                                     pose_cond, inputs["smpl_verts"], inputs["skinning_weights"], inputs["bone_transforms"],
                                     inputs["trans"], inputs["coord_min"], inputs["coord_max"], inputs["center"])
    ws = hip.Workspace(dev)
    tri = torch.from_numpy(g["tri"]).to(dev)
    x_hat = training.unnormalize_canonical_points(tri.reshape(1, -1, 3), inputs["coord_min"][:1], inputs["coord_max"][:1],
                                                  inputs["center"][:1])[0]
    _, x_bar, _ = hip.skin_lbs(frame, ws, x_hat)
    posed = x_bar + inputs["trans"].reshape(1, 3)
    np.testing.assert_allclose(posed.cpu().numpy(), g["posed_verts"], rtol=1e-4, atol=2e-5)
    cam_rot, cam_trans, K = inputs["cam_rot"][0], inputs["cam_trans"][0], inputs["intrinsics"][0]
    p2f = hip.rasterize(meshing.project_opencv(torch.from_numpy(g["posed_verts"]).to(dev).reshape(-1, 3, 3), cam_rot, cam_trans, K), 512, 512)
    assert (p2f.cpu().numpy() == g["p2f_posed"]).all()                # same vertices in: the same pixels out, all of them
    out, _ = meshing.canonical_mesh_outputs(frame, ws, inputs, tri=tri)
    for k in ("output_normal", "normal_cano_front", "normal_cano_back"):
        a, b = out[k].cpu().numpy(), g[k]
        assert a.shape == b.shape == (1, 512, 512, 3)
        same = (np.abs(a - b) <= 2e-4).all(-1)
        assert same.mean() >= 0.9995, (k, same.mean())   # edge pixels: the projections run in the GPU's fp32 (and, posed view, on the build's own skinning)


@gpu
@pytest.mark.parametrize("name,frame_idx", [("zju377_mono", 1), ("zju377_mono", 5), ("h36m", 3)])
def test_band_lattice_gives_the_full_lattices_mesh(scene, name, frame_idx):
    """arah_sdf_grid_band (the 256^3 lattice evaluated only where the level set can pass, csrc/tier.hpp) against arah_sdf_grid:
    the values are EQUAL wherever the band evaluated, the signs are equal everywhere, and marching cubes gives the same triangle
    soup bit for bit -- with a few per cent of the evaluations."""
    from arah_release_amd import hip
    dev = torch.device("cuda:0")
    model, cfg = get_model(name, dev)
    inputs = scene.make_inputs(256, 256, frame_idx=frame_idx, device=dev, max_rays=1024)
    with torch.no_grad():
        model(inputs, eval=True)
    frame, ws = model.idhr_network.last_frame, model.idhr_network.ray_tracer.workspace(dev)
    full = hip.sdf_grid(frame, ws, 256)
    band, n_eval = hip.sdf_grid_band(frame, ws, 256)
    n_eval = int(n_eval.item())
    assert 0 < n_eval < 0.25 * 256 ** 3
    same = band == full
    assert int(same.sum()) >= n_eval                                        # every evaluated point carries the full lattice's value
    assert bool(((band < 0) == (full < 0)).all())                            # the sign pattern marching cubes sees is the full one
    t_full, n_full = hip.marching_cubes(full, 0.0)
    t_band, n_band = hip.marching_cubes(band, 0.0)
    assert int(n_full.item()) == int(n_band.item()) > 1000
    assert torch.equal(t_full, t_band)
    print("band lattice: %d of %d points evaluated (%.1f %%), %d triangles" % (n_eval, 256 ** 3, 100.0 * n_eval / 256 ** 3, int(n_full.item())))
