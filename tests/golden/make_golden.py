"""Generate the golden fixtures in tests/golden/*.npz by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference, see ref_shim.py).  The synthetic subject
(arah_release_amd.synthetic + assets/synthetic_weights.npz) is loaded into the reference's own model
(built through its own ``config.get_model``) under the reference's own parameter names, then the
reference's functions are called directly and their inputs/outputs are stored.

    python tests/golden/make_golden.py            # writes F1..F7 (about 9 MB)
    python tests/golden/make_golden.py f8         # writes F8 (training step, about 3 MB: with strided gradient vectors since round 6)

Fixtures:
  f1_broyden3.npz      broyden() KAT, D=3 (g = LBS(x) - target)
  f1_broyden4.npz      D=4: search_iso_surface_depth (joint residual, Jacobian assembly, broyden) -- `make_golden.py f1d4`
  f2_pointwise.npz     hierarchical_softmax, (un)normalize, skinning, Deformer fwd, query_weights,
                       forward_skinning, forward_skinning_jac
  f3_sdf.npz           emitted SDF MLP: value, feature, autograd gradient
  f4_color_<mode>.npz  RenderingNetwork forward, both colour modes
  f5_tracer_<cfg>.npz  BodyRayTracing.forward 7-tuple (and its inputs), (64,16,16) and (32,8,8)
  f6_shade_<cfg>.npz   get_rbg_value_vol_sdf on the f5 outputs
  f7_forward_<cfg>.npz whole MetaAvatarRender.forward(eval=True) dict for small frames
  f8_train_step_*.npz  training forward + IDHRLoss + backward: outputs, loss terms, per-parameter gradient norms,
                       and the reference's recorded torch.rand draws
  f19_jitter_depths.npz the training-time depth sampler (perturb_z_vals / ray_sampler's z_vals) with its draws -- `make_golden.py f19`
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_shim  # noqa: E402

ref_shim.install()
os.chdir(ref_shim.REF_ROOT)

import im2mesh.config as ref_config  # noqa: E402
from im2mesh.metaavatar_render import config as ref_render_config  # noqa: E402
from im2mesh.utils import root_finding_utils as RFU  # noqa: E402
from im2mesh.utils.broyden import broyden as ref_broyden  # noqa: E402
from im2mesh.utils.utils import hierarchical_softmax as ref_hsoftmax  # noqa: E402
from im2mesh.utils import diff_operators  # noqa: E402

from arah_release_amd import config as my_config  # noqa: E402
from arah_release_amd import synthetic  # noqa: E402

REF_CFG = {"zju377_mono": "configs/arah-zju/ZJUMOCAP-377-mono_4gpus.yaml",
           "zju313": "configs/arah-zju/ZJUMOCAP-313_4gpus.yaml",
           "h36m": "configs/arah-h36m/H36M_S9_4gpus.yaml"}


def build_reference_model(name, n_steps=64, near=16, far=16):
    cfg = ref_config.load_config(REF_CFG[name], "configs/default.yaml")
    cfg["model"].update(n_steps=n_steps, near_surface_samples=near, far_surface_samples=far)
    my_cfg = my_config.builtin_config(name, n_steps, near, far)
    sd = my_config.synthetic_state_dict(my_cfg)
    fake = "/tmp/arah_fake_ckpt_%s.ckpt" % name
    torch.save({"state_dict": {"model.latent.weight": sd["latent.weight"]}}, fake)
    torch.manual_seed(0)
    model = ref_render_config.get_model(cfg, mode="test", checkpoint_path=fake)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys   # our names ARE the reference's names
    # shapes restated in builtin_config must equal the reference yaml's
    for k in ("renderer_kwargs", "decoder_kwargs", "skinning_decoder_kwargs", "cano_view_dirs",
              "color_pose_encoder", "geo_pose_encoder"):
        assert cfg["model"][k] == my_cfg["model"][k], (k, cfg["model"][k], my_cfg["model"][k])
    return model.eval(), cfg


def npify(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        elif isinstance(v, (np.ndarray, float, int, bool, np.floating, np.integer)):
            out[k] = np.asarray(v)
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **npify(arrays))
    print("wrote %-28s %.2f MB" % (name, os.path.getsize(path) / 1e6))


def frame_tensors(model, inputs):
    """What the reference's tracer / renderer need besides the rays."""
    dec_in = {"coords": torch.zeros(1, 1, 3), "rots": inputs["rots"][:1], "Jtrs": inputs["Jtrs"][:1],
              "latent": model.latent(inputs["geo_latent_code_idx"])}
    with torch.no_grad():
        sdf_network = model.sdf_decoder(dec_in)["decoder"]
    B = inputs["rots"].shape[0]
    return dict(sdf_network=sdf_network, loc=torch.zeros(B, 1, 3), sc_factor=torch.ones(B, 1, 1),
                vol_feat=torch.empty(B, 0))


def make_f8():
    """F8: one training step (forward, loss, backward) of the reference on 2048 synthetic rays, ZJUMOCAP-313 shapes
    (idr colour net, cano_view_dirs False, train_skinning_net True) with a fixed view-rotation augmentation.
    The reference's torch.rand draws are recorded so that the build can replay them."""
    from im2mesh.metaavatar_render.renderer.loss import IDHRLoss
    torch.set_num_threads(os.cpu_count())
    scene = synthetic.SyntheticScene(seed=0)
    cfg = ref_config.load_config(REF_CFG["zju313"], "configs/default.yaml")
    cfg["training"].update(pose_input_noise=False, view_input_noise=False)   # the np.random gate is not replayable
    my_cfg = my_config.builtin_config("zju313")
    sd = my_config.synthetic_state_dict(my_cfg)
    fake = "/tmp/arah_fake_ckpt_f8.ckpt"
    torch.save({"state_dict": {"model.latent.weight": sd["latent.weight"]}}, fake)
    torch.manual_seed(0)
    model = ref_render_config.get_model(cfg, mode="test", checkpoint_path=fake)
    model.load_state_dict(sd, strict=False)
    model.train()
    for k in ("rgb_weight", "perceptual_weight", "eikonal_weight", "mask_weight", "off_surface_weight", "inside_weight",
              "params_weight", "skinning_weight", "rgb_loss_type"):
        assert cfg["training"][k] == my_cfg["training"][k], (k, cfg["training"][k], my_cfg["training"][k])
    inputs = scene.make_inputs(128, 128, frame_idx=2, max_rays=2048, eval_mode=False)
    ang = np.deg2rad(15.0)
    view_noise = torch.tensor([[[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]], dtype=torch.float32)
    inputs["pose_cond"]["view_noise"] = view_noise
    draws, orig = [], torch.rand

    def recording_rand(*a, **k):
        t = orig(*a, **k)
        draws.append(t.detach().clone())
        return t

    torch.manual_seed(1234)
    torch.rand = recording_rand
    try:
        out = model(inputs)
    finally:
        torch.rand = orig
    assert [tuple(d.shape) for d in draws] == [(1, 2048, 64), (1, 2048, 17), (1, 2048, 16), (1, 1024, 3)], \
        [tuple(d.shape) for d in draws]
    t = cfg["training"]
    crit = IDHRLoss(t["rgb_weight"], t["perceptual_weight"], t["eikonal_weight"], t["mask_weight"], t["off_surface_weight"],
                    t["inside_weight"], t["params_weight"], t["skinning_weight"], t["rgb_loss_type"])
    loss = crit(out, {"rgb": inputs["rgb_values"], "sampled_weights": inputs["sampled_weights"]})
    loss["loss"].backward()
    grads = {"grad." + n: p.grad.norm() for n, p in model.named_parameters() if p.grad is not None}
    n_no_grad = sum(1 for n, p in model.named_parameters() if p.grad is None)
    print("parameters with / without gradient:", len(grads), n_no_grad)
    # round 6: gradient DIRECTIONS, not only norms.  Every tensor's gradient, strided so that the fixture stays ~2 MB:
    # whole tensors for beta, the latent code, the skinning MLP, the pose encoder and the colour MLP's gains and biases; every 4th
    # element of the FiLM mapping network, every 8th of the colour MLP's weight_v, every 509th / 13th of the hypernetwork's large /
    # medium tensors (its small ones whole).
    # ("gvec.<name>" = grad.reshape(-1)[::stride], "gstride.<name>" = stride)
    def stride_of(n, p):
        if n.startswith("sdf_decoder.net.layers"):
            return 509 if p.numel() >= (1 << 20) else (13 if p.numel() >= (1 << 14) else 1)   # primes: a power of two walks one column
        if n.startswith("sdf_decoder.net.mapping_network"):
            return 4
        if n.startswith("color_decoder") and n.endswith("weight_v"):
            return 8
        return 1
    for n, p in model.named_parameters():
        if p.grad is not None:
            st = stride_of(n, p)
            grads["gvec." + n] = p.grad.reshape(-1)[::st].clone()
            grads["gstride." + n] = torch.tensor(st)
    save("f8_train_step_zju313.npz", frame_idx=2, H=128, W=128, max_rays=2048, view_noise=view_noise,
         rand_steps=draws[0][0], rand_near=draws[1][0], rand_far=draws[2][0], rand_eikonal=draws[3][0],
         rgb_values=out["rgb_values"][0], sdf_output=out["sdf_output"][0], network_body_mask=out["network_body_mask"][0],
         off_surface_sdf=out["off_surface_sdf"][0], grad_theta=out["grad_theta"], pred_weights=out["pred_weights"][0],
         inside_sdf=out["inside_sdf"], **{"loss." + k: v.reshape(-1)[0] for k, v in loss.items()}, **grads)


def make_f1_d4():
    """F1, D = 4: the reference's joint root find (search_iso_surface_depth: Jacobian assembly, residual closure and
    broyden() on u = (x_hat, depth), root_finding_utils.py:365-484) on 256 rays with perturbed starts."""
    torch.set_num_threads(os.cpu_count())
    scene = synthetic.SyntheticScene(seed=0)
    model, cfg = build_reference_model("zju377_mono")
    inputs = scene.make_inputs(64, 64, frame_idx=0)
    ft = frame_tensors(model, inputs)
    sdf_network, loc, sc, vol = ft["sdf_network"], ft["loc"], ft["sc_factor"], ft["vol_feat"]
    cmin, cmax, center = inputs["coord_min"], inputs["coord_max"], inputs["center"]
    bones, trans = inputs["bone_transforms"], inputs["trans"]
    skin = model.skinning_model
    g = torch.Generator().manual_seed(321)
    # canonical points close to the zero level set: project random points with a few Newton steps on the SDF
    P = 256
    x_norm = (torch.rand(1, 4096, 3, generator=g) * 1.4 - 0.7)
    for _ in range(6):
        with torch.enable_grad():
            xg = x_norm.clone().requires_grad_(True)
            sdf = sdf_network(xg)
            grad = diff_operators.gradient(sdf, xg, create_graph=False)
        x_norm = (xg - sdf * grad / (grad.pow(2).sum(-1, keepdim=True) + 1e-8)).detach()
    with torch.no_grad():
        keep = sdf_network(x_norm)[0, :, 0].abs().argsort()[:P]
        x_norm = x_norm[:, keep]
        x_hat = RFU.unnormalize_canonical_points(x_norm, cmin, cmax, center)
        x_bar, T_true = RFU.forward_skinning(x_hat, loc, sc, cmin, cmax, center, skin, vol, bones)
        x_posed = x_bar + trans
        cam = torch.zeros(1, P, 3)
        depth = x_posed.norm(dim=-1)
        rays = x_posed / depth[..., None]
        # perturbed starts; a few hopeless ones exercise the divergence / best-iterate paths; some rays masked out
        x0 = x_hat + torch.randn(1, P, 3, generator=g) * 0.01
        z0 = depth + torch.randn(1, P, generator=g) * 0.01
        x0[:, :6] += 0.4
        valid = torch.ones(1, P, dtype=torch.bool)
        valid[:, 10:20] = False
        T0 = T_true + torch.randn(1, P, 4, 4, generator=g) * 1e-3
        xo, zo, To, conv = RFU.search_iso_surface_depth(cam, rays, valid, x0, z0, T0, sdf_network, loc, sc, skin, vol,
                                                        bones, trans, cmin, cmax, center, eval_mode=True)
    save("f1_broyden4.npz", cam=cam[0], rays=rays[0], valid=valid[0], x0=x0[0], z0=z0[0], T0=T0[0], x_opt=xo[0],
         z_opt=zo[0], T_opt=To[0], converged=conv[0])


def make_f9():
    """F9: the callers' side -- SMPL LBS (human_body_prior.lbs), Vitruvian-pose transforms (torch and numpy twins),
    box / ray helpers, and LightningModel.compose_inputs in both branches (dataset-provided SMPL; optimised SMPL +
    cameras), on the synthetic body model.  The dataset's __getitem__ itself needs cv2.fillPoly and cannot run here:
    the data dict comes from the build's device-side counterpart (arah_release_amd.data.frame_item) and is stored."""
    import torch.nn as nn
    from human_body_prior.body_model.lbs import lbs as ref_lbs
    from im2mesh.metaavatar_render import lightning_model as ref_lm
    from im2mesh.metaavatar_render import models as ref_models
    from im2mesh.data.zju_mocap_odp import get_02v_bone_transforms
    from im2mesh.utils.utils import get_near_far
    from scipy.spatial.transform import Rotation
    from arah_release_amd import data as my_data, smpl as my_smpl
    torch.set_num_threads(os.cpu_count())
    scene = synthetic.SyntheticScene(seed=0)
    body = my_smpl.BodyModel.synthetic(scene)
    rng = np.random.RandomState(9)
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    out = {}
    # ---- lbs
    betas = t(rng.randn(1, 10) * 0.5)
    pose = t(rng.randn(1, 72) * 0.3)
    verts, J_t, J, A, abs_A, v_posed = ref_lbs(betas=betas, pose=pose, v_template=t(body.v_template)[None],
                                               clothed_v_template=None, shapedirs=t(body.shapedirs), posedirs=t(body.posedirs),
                                               J_regressor=t(body.J_regressor),
                                               parents=torch.from_numpy(body.kintree_table[0].astype(np.int64)),
                                               lbs_weights=t(body.lbs_weights), dtype=torch.float32)
    out.update(lbs_betas=betas, lbs_pose=pose, lbs_verts=verts[0], lbs_J_posed=J_t[0], lbs_J=J[0], lbs_A=A[0],
               lbs_v_posed=v_posed[0])
    # ---- 02v transforms
    out.update(v02_torch=ref_lm.get_transforms_02v(J[0]),
               v02_numpy=get_02v_bone_transforms(J[0].numpy(), Rotation.from_euler("z", 45, degrees=True).as_matrix(),
                                                 Rotation.from_euler("z", -45, degrees=True).as_matrix()))
    # ---- rays / near-far
    bounds = np.array([[-0.4, -0.9, 2.6], [0.5, 0.8, 3.3]], np.float32)
    ray_d = rng.randn(512, 3).astype(np.float32)
    ray_d[:, 2] = np.abs(ray_d[:, 2]) + 1.0
    ray_d[:8, 0] = 0.0
    ray_o = np.broadcast_to(np.array([0.02, -0.01, 0.0], np.float32), ray_d.shape)
    near, far, ok = get_near_far(bounds, ray_o, ray_d.copy())
    Rm = Rotation.from_rotvec([0.1, -0.2, 0.05]).as_matrix().astype(np.float32)
    uv = rng.randn(1, 256, 3).astype(np.float32)
    out.update(nf_bounds=bounds, nf_ray_o=ray_o[:1], nf_ray_d=ray_d, nf_near=near, nf_far=far, nf_ok=ok, cam_R=Rm,
               cam_uv=uv[0], cam_rays=ref_lm.get_camera_rays(t(Rm)[None], t(uv))[0],
               cam_loc=ref_lm.get_camera_location(t(Rm)[None], t([[0.3, -0.1, 2.0]]))[0])
    # ---- compose_inputs
    aa = rng.randn(24, 3) * 0.25
    Aposed = scene.frame(3)
    model_dict = dict(minimal_shape=scene.verts_cano, betas=np.zeros((1, 10), np.float32), Jtr_posed=Aposed["joints_posed"],
                      bone_transforms=Aposed["bone_transforms"], trans=np.array([0.1, 0.0, 3.0], np.float32),
                      root_orient=aa[0], pose_body=aa[1:22].reshape(-1), pose_hand=aa[22:].reshape(-1))
    model_dict = {k: np.asarray(v, np.float32) for k, v in model_dict.items()}
    cam = {"K": np.array([[76.8, 0, 32], [0, 76.8, 32], [0, 0, 1]], np.float32), "R": Rotation.from_rotvec([0.0, 0.05, 0.0]).as_matrix(),
           "T": np.array([0.02, 0.0, 0.1], np.float32)}
    item = my_data.frame_item(model_dict, cam, body, 64, 64, frame_idx=5, data_idx=1)
    cfg = ref_config.load_config(REF_CFG["zju313"], "configs/default.yaml")
    my_cfg = my_config.builtin_config("zju313")
    sd = my_config.synthetic_state_dict(my_cfg)
    fake = "/tmp/arah_fake_ckpt_f9.ckpt"
    torch.save({"state_dict": {"model.latent.weight": sd["latent.weight"]}}, fake)
    model = ref_render_config.get_model(cfg, mode="test", checkpoint_path=fake)
    lm = ref_lm.LightningModel.__new__(ref_lm.LightningModel)
    nn.Module.__init__(lm)
    lm.model, lm.cfg = model, cfg
    object.__setattr__(lm, "device", torch.device("cpu"))
    model.frames = []
    a = lm.compose_inputs(dict(item), eval=True)
    keep = ("ray_dirs", "cam_loc", "pose", "bone_transforms", "trans", "coord_min", "coord_max", "center", "Jtrs", "rots",
            "smpl_verts", "minimal_shape", "cam_rot", "cam_trans")
    out.update({"ciA." + k: a[k] for k in keep})
    out.update({"ciA.rots_full": a["pose_cond"]["rots_full"], "ciA.Jtrs_posed": a["pose_cond"]["Jtrs_posed"],
                "ciA.latent_code_idx": a["pose_cond"]["latent_code_idx"], "ciA.geo_latent_code_idx": a["geo_latent_code_idx"]})
    # branch B: what MetaAvatarRender.__init__ registers with train_smpl / train_cameras (models/__init__.py:81-123);
    # the constructor itself reads body_models/misc/*.npz, which do not exist here
    model.train_smpl = model.train_cameras = True
    model.frames = [4, 5]
    for name in ("posedirs", "shapedirs", "J_regressor", "lbs_weights"):
        model.register_buffer(name, t(getattr(body, name)))
    model.register_buffer("v_template", t(body.v_template).unsqueeze(0))
    model.register_buffer("kintree_table", torch.from_numpy(body.kintree_table.astype(np.int32)))
    pd = {}
    smpl_b = {}
    for fr in (4, 5):
        p = rng.randn(24, 3).astype(np.float32) * 0.2
        smpl_b[fr] = dict(root_orient=p[0], pose_body=p[1:22].reshape(-1), pose_hand=p[22:].reshape(-1),
                          trans=np.array([0.05 * fr, 0.01, 3.0], np.float32))
        pd.update({"%s_%d" % (k, fr): nn.Parameter(t(v)) for k, v in smpl_b[fr].items()})
    model.body_poses = nn.ParameterDict(pd)
    model.register_parameter("betas", nn.Parameter(t(rng.randn(1, 10) * 0.3)))
    quat = Rotation.from_rotvec([[0.02, -0.04, 0.01]]).as_quat().astype(np.float32)
    model.register_parameter("cam_rots", nn.Parameter(t(quat)))
    model.register_parameter("cam_trans", nn.Parameter(t([[0.01, 0.02, 0.12]])))
    itemB = dict(item)
    itemB["inputs.novel_seq"] = None
    itemB["inputs"] = torch.rand(1, item["inputs.uv"].shape[1], 3)
    b = lm.compose_inputs(itemB, eval=False)
    out.update({"ciB." + k: b[k] for k in keep})
    out.update({"ciB.rots_full": b["pose_cond"]["rots_full"], "ciB.Jtrs_posed": b["pose_cond"]["Jtrs_posed"],
                "ciB.latent_code_idx": b["pose_cond"]["latent_code_idx"], "ciB.betas": model.betas, "ciB.cam_rots": model.cam_rots,
                "ciB.cam_trans": model.cam_trans, "ciB.rgb_values": b["rgb_values"]})
    for fr in (4, 5):
        out.update({"ciB.%s_%d" % (k, fr): v for k, v in smpl_b[fr].items()})
    out.update({"md." + k: v for k, v in model_dict.items()})
    out.update({"cam." + k: np.asarray(v, np.float32) for k, v in cam.items()})
    save("f9_callers.npz", **out)


def test_mesh():
    """A closed, body-sized triangle mesh for the mesh-query fixtures: zero level of a union of capsules (torso, head,
    two arms, two legs) on a 56^3 lattice, triangulated by the build's own marching cubes (CPU tensors)."""
    from arah_release_amd import meshing
    n = 56
    lin = torch.linspace(-1.0, 1.0, n)
    X, Y, Z = torch.meshgrid(lin, lin, lin, indexing="ij")
    P = torch.stack([X, Y, Z], -1).reshape(-1, 3)

    def capsule(a, b, r):
        a, b = torch.tensor(a), torch.tensor(b)
        ab = b - a
        t = ((P - a) @ ab / (ab @ ab)).clamp(0, 1)
        return (P - a - t[:, None] * ab).norm(dim=-1) - r

    parts = [capsule((0, -0.1, 0), (0, 0.45, 0), 0.2), capsule((0, 0.62, 0), (0, 0.7, 0), 0.12),
             capsule((-0.22, 0.4, 0), (-0.75, 0.42, 0.05), 0.07), capsule((0.22, 0.4, 0), (0.75, 0.38, -0.05), 0.07),
             capsule((-0.1, -0.2, 0), (-0.18, -0.9, 0.03), 0.09), capsule((0.1, -0.2, 0), (0.2, -0.9, -0.02), 0.09)]
    sdf = torch.stack(parts, 0).min(0).values.reshape(n, n, n)
    tri = meshing.marching_cubes(sdf, level=0.0).float()                         # (F,3,3) soup, shared corners bit-equal
    verts, inv = torch.unique(tri.reshape(-1, 3), dim=0, return_inverse=True)
    return verts.numpy().astype(np.float32), inv.reshape(-1, 3).numpy().astype(np.int32)


def make_f10():
    """F10: containment of query points in a closed mesh by the reference's OWN im2mesh/utils/libmesh/inside_mesh.py
    (check_mesh_contains, zju_mocap.py:466,493,520), its Cython triangle hash replaced by a brute-force candidate filter
    (ref_shim._BruteTriangleHash: the hash is only a pre-selection).  The mesh object is a stand-in for trimesh.Trimesh
    (float64 ``vertices``, ``faces``)."""
    import types
    from im2mesh.utils.libmesh.inside_mesh import check_mesh_contains
    verts, faces = test_mesh()
    rng = np.random.RandomState(7)
    uniform = rng.rand(4096, 3) * 2.0 - 1.0                                   # zju_mocap.py:464
    fa = rng.randint(0, len(faces), 2048)
    w = rng.dirichlet(np.ones(3), 2048)
    surf = (verts[faces[fa]].astype(np.float64) * w[..., None]).sum(1)
    near = surf + rng.normal(scale=0.02, size=surf.shape)                     # both sides of the surface, 2 cm
    far = surf + rng.normal(scale=0.5, size=surf.shape)                       # zju_mocap.py:518
    pts = np.concatenate([uniform, near, far, verts[:64].astype(np.float64)], 0)
    mesh = types.SimpleNamespace(vertices=verts.astype(np.float64), faces=faces.astype(np.int64))
    contains = check_mesh_contains(mesh, pts)
    np.savez_compressed(os.path.join(HERE, "f10_mesh_contains.npz"), verts=verts, faces=faces, points=pts, contains=contains)
    print("wrote f10_mesh_contains.npz: %d verts, %d faces, %d points, %d inside" % (len(verts), len(faces), len(pts), int(contains.sum())))


def make_f11():
    """F11: the optimiser LightningModel.configure_optimizers builds (lightning_model.py:403-461): per parameter group
    its learning rate, weight decay, number of tensors and number of elements, in order -- plain ZJUMOCAP-313, and with
    optimised SMPL parameters and cameras (registered as in make_f9: the constructor needs the SMPL files)."""
    import json
    import torch.nn as nn
    from im2mesh.metaavatar_render import lightning_model as ref_lm
    cfg = ref_config.load_config(REF_CFG["zju313"], "configs/default.yaml")
    sd = my_config.synthetic_state_dict(my_config.builtin_config("zju313"))
    fake = "/tmp/arah_fake_ckpt_f11.ckpt"
    torch.save({"state_dict": {"model.latent.weight": sd["latent.weight"]}}, fake)
    out = {}
    for tag, flags in (("plain", (False, False)), ("smpl_cameras", (True, True))):
        cfg["model"]["train_smpl"], cfg["model"]["train_cameras"] = False, False
        model = ref_render_config.get_model(cfg, mode="test", checkpoint_path=fake)
        if flags[0]:
            model.train_smpl = model.train_cameras = True
            pd = {"%s_%d" % (k, fr): nn.Parameter(torch.zeros(n)) for fr in (4, 5)
                  for k, n in (("root_orient", 3), ("pose_body", 63), ("pose_hand", 6), ("trans", 3))}
            model.body_poses = nn.ParameterDict(pd)
            model.register_parameter("betas", nn.Parameter(torch.zeros(1, 10)))
            model.register_parameter("cam_rots", nn.Parameter(torch.zeros(2, 4)))
            model.register_parameter("cam_trans", nn.Parameter(torch.zeros(2, 3)))
        cfg["model"]["train_smpl"], cfg["model"]["train_cameras"] = flags
        lm = ref_lm.LightningModel.__new__(ref_lm.LightningModel)
        nn.Module.__init__(lm)
        lm.model, lm.cfg = model, cfg
        opt = lm.configure_optimizers()
        out[tag] = [{"lr": g["lr"], "weight_decay": g["weight_decay"], "tensors": len(g["params"]),
                     "elements": int(sum(p.numel() for p in g["params"]))} for g in opt.param_groups]
        out[tag + "_adam"] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in opt.defaults.items()
                              if k in ("lr", "betas", "eps", "weight_decay", "amsgrad")}
    with open(os.path.join(HERE, "f11_optimizer_groups.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote f11_optimizer_groups.json", {k: len(v) for k, v in out.items()})


def make_f12():
    """F12: what the reference's load_config makes of its own YAML files (recursive inherit_from + configs/default.yaml,
    im2mesh/config.py:12-56) for the three configurations the build restates in builtin_config: the merged 'model',
    'training' and 'data' sections as plain JSON."""
    import json
    out = {}
    for name, path in REF_CFG.items():
        cfg = ref_config.load_config(path, "configs/default.yaml")
        out[name] = {sec: cfg.get(sec, {}) for sec in ("method", "model", "training", "data")}
    with open(os.path.join(HERE, "f12_merged_configs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote f12_merged_configs.json", list(out))


def make_f13():
    """F13: the reference's IDHRLoss (renderer/loss.py) on random model outputs -- every term in isolation from the forward
    pass: three pixel-loss types, the boundary value 100 of the body mask, empty hit / off-surface masks, more than 2048
    rays (the patch tail is ignored by every term but the perceptual one, whose weight is 0: LPIPS is not available)."""
    from im2mesh.metaavatar_render.renderer.loss import IDHRLoss
    g = torch.Generator().manual_seed(77)
    cases = {}
    for ci, (kind, n_rays, empty_hit, empty_off, boundary) in enumerate((("l1", 2048, False, False, False),
                                                                           ("mse", 2048, False, False, True),
                                                                           ("smoothed_l1", 2100, False, False, True),
                                                                           ("l1", 512, True, False, False),
                                                                           ("l1", 512, False, True, False))):
        crit = IDHRLoss(1.0, 0.0, 0.1, 0.5, 0.3, 0.2, 1e-3, 10.0, rgb_loss_type=kind, perceptual_loss_fn=None)
        hit = torch.rand(1, n_rays, generator=g) > (2.0 if empty_hit else 0.4)
        body = (torch.rand(1, n_rays, generator=g) > 0.5).long()
        if boundary:
            body[0, ::7] = 100
        off = torch.rand(1, n_rays, generator=g) > (2.0 if empty_off else 0.6)
        out = {"rgb_values": torch.rand(1, n_rays, 3, generator=g), "network_body_mask": hit, "body_mask": body,
               "off_surface_mask": off, "sdf_output": torch.rand(1, min(n_rays, 2048), 1, generator=g),   # the mask term indexes it with a 2048-ray mask
               "grad_theta": torch.randn(1500, 3, generator=g), "off_surface_sdf": torch.rand(1, 1024, 1, generator=g) * 0.1,
               "inside_sdf": torch.randn(1024, 1, generator=g) * 1e-3,
               "sdf_params": [torch.randn(1, 1000 + 10 * k, generator=g) * 0.01 for k in range(3)],
               "pred_weights": torch.softmax(torch.randn(1, 1024, 24, generator=g), -1),
               "surface_normals": None}          # read by the reference's forward, used by no term
        gt = {"rgb": torch.rand(1, n_rays, 3, generator=g), "sampled_weights": torch.softmax(torch.randn(1, 1024, 24, generator=g), -1)}
        with torch.no_grad():
            res = crit(out, gt)
        rec = {"kind": np.array(kind), "n_sdf_params": np.array(3)}
        rec.update({"out." + k: v.numpy() for k, v in out.items() if torch.is_tensor(v)})
        rec.update({"out.sdf_params_%d" % k: v.numpy() for k, v in enumerate(out["sdf_params"])})
        rec.update({"gt." + k: v.numpy() for k, v in gt.items()})
        rec.update({"res." + k: np.asarray(v.detach().numpy(), np.float64) for k, v in res.items()})
        cases.update({"c%d.%s" % (ci, k): v for k, v in rec.items()})
    np.savez_compressed(os.path.join(HERE, "f13_idhr_loss.npz"), **cases)
    print("wrote f13_idhr_loss.npz", sorted({k.split(".", 1)[1] for k in cases if k.startswith("c0.res.")}))


def make_f14():
    """F14: the reference's LightningModel.validation_step (lightning_model.py:160-230) around a stub model: scatter of the
    rendered pixels into the image, the normal map it derives from the camera-space surface points by finite differences
    (incl. its NaN handling at the silhouette), the ground-truth image and PSNR.  SSIM and LPIPS (skimage, lpips: absent)
    are replaced by constants."""
    import torch.nn as nn
    from im2mesh.metaavatar_render import lightning_model as ref_lm
    ref_lm.ssim_metric = lambda *a, **k: 0.5
    ref_lm.lpips_metric = lambda *a, **k: 0.25
    g = torch.Generator().manual_seed(14)
    H, W = 24, 20
    mask = torch.zeros(1, H, W, dtype=torch.bool)
    mask[0, 3:21, 4:17] = True
    mask[0, 10, 9] = False
    n = int(mask.sum())
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    depth = 3.0 + 0.2 * torch.sin(xx / 3.0) * torch.cos(yy / 4.0)
    pts_img = torch.stack([(xx - 10) / 40.0 * depth, (yy - 12) / 40.0 * depth, depth], -1)
    hit = torch.rand(H, W, generator=g) > 0.25                      # rays that missed the body: points_cam == 0
    pts = (pts_img * hit[..., None])[mask[0]]
    outputs = {"rgb_values": torch.rand(1, n, 3, generator=g), "points_cam": pts.unsqueeze(0)}

    class Stub(nn.Module):
        def forward(self, inputs, gen_cano_mesh=False, eval=True):
            return dict(outputs)

    lm = ref_lm.LightningModel.__new__(ref_lm.LightningModel)
    nn.Module.__init__(lm)
    lm.model, lm.cfg, lm.loss_fn_vgg = Stub(), {}, None
    object.__setattr__(lm, "device", torch.device("cpu"))
    lm.compose_inputs = lambda batch, eval: {}
    batch = {"inputs.img_height": torch.tensor([H]), "inputs.img_width": torch.tensor([W]), "inputs.image_mask": mask,
             "inputs": torch.rand(1, n, 3, generator=g)}
    res = lm.validation_step(batch, 0)
    save("f14_validation_step.npz", H=H, W=W, image_mask=mask, rgb_values=outputs["rgb_values"], points_cam=outputs["points_cam"],
         inputs=batch["inputs"], psnr=np.float64(res["psnr"]), ssim=np.float64(res["ssim"]), lpips=np.float64(res["lpips"]),
         rgb_pred=res["rgb_pred"], normal_pred=res["normal_pred"], rgb_gt=res["rgb_gt"])
    print("psnr", res["psnr"], "nan->-1 pixels", int((res["normal_pred"] == 0).all(0).sum()))


def make_f7(which, full_frames=False):
    """F7: the reference's whole MetaAvatarRender.forward(eval=True) on small synthetic frames.  The last entry is BASELINE
    config 5's sampling (128 samples per ray, 32 near / 32 far, H36M shapes) on a small frame.
    full_frames (round 5, `f7full`): BASELINE configs 1 and 2 at their FULL sizes -- 256 x 256 x 32 and the benchmark frame
    itself, 512 x 512 x 64 -- rendered by the reference on the CPU (one and nine minutes on eight cores); the rays are what
    synthetic.SyntheticScene makes for the frame and are not stored again."""
    import time
    torch.set_num_threads(os.cpu_count())
    scene = synthetic.SyntheticScene(seed=0)
    for name, (H, W), (S, near, far), fidx in which:
        model, cfg = build_reference_model(name, S, near, far)
        inputs = scene.make_inputs(H, W, frame_idx=fidx)
        t0 = time.time()
        with torch.no_grad():
            out = model(inputs, gen_cano_mesh=False, eval=True)
        extra = {} if full_frames else dict(ray_dirs=inputs["ray_dirs"][0],
                                            body_bounds_intersections=inputs["body_bounds_intersections"][0])
        if full_frames:
            extra["reference_seconds"] = np.float64(time.time() - t0)
            extra["reference_threads"] = np.int64(torch.get_num_threads())
        save("f7_forward_%s_%dx%d_s%d.npz" % (name, H, W, S), frame_idx=fidx, H=H, W=W, n_steps=S, n_near=near,
             n_far=far, rgb_values=out["rgb_values"][0], points_cam=out["points_cam"][0],
             network_body_mask=out["network_body_mask"][0], sdf_param0=out["sdf_params"][0][0, :16], **extra)


def make_f15():
    """F15: the lattice of the canonical-mesh branch as the reference's OWN create_mesh_vertices_and_faces
    (utils/sdf_meshing.py:13-70) builds it -- the only parts of that branch that are reference code rather than skimage /
    pytorch3d: which coordinates the decoder is asked for and in which order, how the values are laid out in the (N,N,N)
    volume handed to marching cubes, the level / spacing it is called with, and the map from its vertices back to
    coordinates.  `.cuda()` is patched to the identity, the decoder records its inputs and returns an asymmetric affine
    function of them, skimage's marching_cubes_lewiner is replaced by a recorder that returns a fixed vertex list."""
    import skimage.measure as skm
    from im2mesh.utils import sdf_meshing
    rec = {}

    class Decoder(torch.nn.Module):
        def forward(self, x):
            rec.setdefault("coords", []).append(x.reshape(-1, 3).clone())
            rec.setdefault("batch", []).append(x.shape[1])
            return (0.3 * x[..., 0] - 0.5 * x[..., 1] + 0.7 * x[..., 2] + 0.11).unsqueeze(-1)

    def fake_mc(volume, level=None, spacing=None, **kw):
        rec["volume"], rec["level"], rec["spacing"] = np.array(volume), level, np.array(spacing, np.float64)
        idx = np.array([[0.0, 0.0, 0.0], [1.5, 2.25, 3.0], [5.0, 4.0, 0.5], [2.0, 2.0, 2.0]])
        rec["verts_idx"] = idx
        return idx * np.asarray(spacing), np.array([[0, 1, 2], [1, 2, 3]]), np.zeros((4, 3)), np.zeros(4)

    old_cuda, old_mc = torch.Tensor.cuda, getattr(skm, "marching_cubes_lewiner", None)
    torch.Tensor.cuda = lambda self, *a, **k: self
    skm.marching_cubes_lewiner = fake_mc
    try:
        out = {}
        N = 6
        pts, faces = sdf_meshing.create_mesh_vertices_and_faces(Decoder(), N=N, max_batch=50)
        out.update(N=N, coords=torch.cat(rec["coords"]).numpy(), batches=np.array(rec["batch"]), volume=rec["volume"],
                   level=np.float64(rec["level"]), spacing=rec["spacing"], verts_idx=rec["verts_idx"], mesh_points=pts,
                   faces=faces)
        pts2, _ = sdf_meshing.convert_sdf_samples_to_vertices_and_faces(torch.zeros(N, N, N), [-1, -1, -1], 2.0 / (N - 1),
                                                                        offset=np.array([0.1, -0.2, 0.3]), scale=2.0)
        out["mesh_points_scaled"] = pts2
        rec.clear()
        N = 256                                            # the size the model asks for (models/__init__.py:205): one axis
        pts, _ = sdf_meshing.create_mesh_vertices_and_faces(Decoder(), N=N, max_batch=64 ** 3)
        c = torch.cat(rec["coords"]).reshape(N, N, N, 3)
        assert torch.equal(c[:, 0, 0, 0], c[:, 5, 9, 0]) and torch.equal(c[0, :, 0, 1], c[3, :, 8, 1])
        out.update(axis256=c[:, 0, 0, 0].numpy(), axis256_y=c[0, :, 0, 1].numpy(), axis256_z=c[0, 0, :, 2].numpy(),
                   batches256=np.array(rec["batch"]), spacing256=rec["spacing"], mesh_points256=pts)
    finally:
        torch.Tensor.cuda = old_cuda
        if old_mc is not None:
            skm.marching_cubes_lewiner = old_mc
    save("f15_sdf_lattice.npz", **out)


def make_f16():
    """F16: files written by the reference's OWN preprocessing script (preprocess_datasets/preprocess_ZJU-MoCap.py, run as
    __main__): models/000000.npz through its np.savez call (:150-158) and cam_params.json through its json.dump (:163-164).
    The script needs the licensed SMPL model, EasyMocap and a raw capture; here its inputs are synthetic (a one-frame,
    23-camera 'CoreView_377' with random SMPL parameters) and the three things it imports for the body -- human_body_prior's
    BodyModel, EasyMocap's load_model, `.cuda()` -- are stand-ins that return tensors of the real shapes.  What the fixture
    pins is therefore the on-disk SCHEMA as reference code writes it: key names, shapes, dtypes, units (T in metres),
    nesting of the JSON -- what data.load_model_npz / load_cam_params must read."""
    import json
    import runpy
    import shutil
    import tempfile
    import types
    from scipy.spatial.transform import Rotation
    rng = np.random.RandomState(16)
    tmp = tempfile.mkdtemp()
    raw, outd = os.path.join(tmp, "raw", "CoreView_377"), os.path.join(tmp, "out")
    os.makedirs(os.path.join(raw, "Camera_B1"))
    os.makedirs(os.path.join(raw, "mask_cihp", "Camera_B1"))
    os.makedirs(os.path.join(raw, "new_params"))
    os.makedirs(os.path.join(tmp, "body_models", "misc"))
    np.savez(os.path.join(tmp, "body_models", "misc", "faces.npz"), faces=np.zeros((13776, 3), np.int64))
    for f in (os.path.join(raw, "Camera_B1", "000000.jpg"), os.path.join(raw, "mask_cihp", "Camera_B1", "000000.png")):
        open(f, "wb").write(b"x")
    cams = {"K": [np.array([[1000.0 + c, 0, 512], [0, 1001.0 + c, 513], [0, 0, 1]]) for c in range(23)],
            "D": [rng.randn(5, 1) * 0.01 for _ in range(23)], "R": [Rotation.from_rotvec(rng.randn(3) * 0.3).as_matrix() for _ in range(23)],
            "T": [rng.randn(3, 1) * 1000.0 for _ in range(23)]}
    np.save(os.path.join(raw, "annots.npy"), {"cams": cams}, allow_pickle=True)
    np.save(os.path.join(raw, "new_params", "0.npy"), {"Rh": rng.randn(1, 3) * 0.2, "Th": rng.randn(1, 3), "shapes": rng.randn(1, 10) * 0.5,
                                                      "poses": rng.randn(1, 72) * 0.2}, allow_pickle=True)
    verts = torch.from_numpy(rng.randn(1, 6890, 3).astype(np.float32) * 0.3)

    class _Body:
        v, Jtr = verts, torch.from_numpy(rng.randn(1, 24, 3).astype(np.float32))
        bone_transforms = torch.from_numpy(rng.randn(1, 24, 4, 4).astype(np.float32))

    class BodyModel:
        def __init__(self, *a, **k):
            pass

        def cuda(self):
            return self

        def __call__(self, **k):
            return _Body()

    stubs = {"preprocess_datasets": types.ModuleType("preprocess_datasets"),
             "preprocess_datasets.easymocap": types.ModuleType("easymocap"),
             "preprocess_datasets.easymocap.mytools": types.ModuleType("mytools"),
             "preprocess_datasets.easymocap.mytools.camera_utils": types.ModuleType("camera_utils"),
             "preprocess_datasets.easymocap.smplmodel": types.ModuleType("smplmodel"),
             "human_body_prior.body_model.body_model": types.ModuleType("body_model")}
    for name, mod in stubs.items():          # parents expose their children, packages have a __path__
        parent, _, child = name.rpartition(".")
        if parent in stubs:
            setattr(stubs[parent], child, mod)
            stubs[parent].__path__ = []
    stubs["human_body_prior.body_model.body_model"].BodyModel = BodyModel
    stubs["preprocess_datasets.easymocap.smplmodel"].load_model = lambda **k: (lambda **kk: [verts[0] + 0.01])
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    old_cuda, old_argv, old_cwd = torch.Tensor.cuda, sys.argv, os.getcwd()
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        os.chdir(tmp)
        sys.argv = ["preprocess_ZJU-MoCap.py", "--data-dir", os.path.join(tmp, "raw"), "--out-dir", outd, "--seqname", "CoreView_377"]
        runpy.run_path(os.path.join(ref_shim.REF_ROOT, "preprocess_datasets", "preprocess_ZJU-MoCap.py"), run_name="__main__")
    finally:
        os.chdir(old_cwd)
        sys.argv, torch.Tensor.cuda = old_argv, old_cuda
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    dst = os.path.join(HERE, "f16_preprocessed_CoreView_377")
    os.makedirs(os.path.join(dst, "models"), exist_ok=True)
    shutil.copy(os.path.join(outd, "CoreView_377", "models", "000000.npz"), os.path.join(dst, "models", "000000.npz"))
    shutil.copy(os.path.join(outd, "CoreView_377", "cam_params.json"), os.path.join(dst, "cam_params.json"))
    # what went in, for the reader test
    np.savez(os.path.join(dst, "inputs.npz"), K=np.stack(cams["K"]), D=np.stack(cams["D"]), R=np.stack(cams["R"]), T=np.stack(cams["T"]),
             verts=verts[0].numpy(), Jtr=_Body.Jtr[0].numpy(), bone_transforms=_Body.bone_transforms[0].numpy())
    shutil.rmtree(tmp)
    print("wrote", dst, os.listdir(dst))


def make_f17():
    """F17: the pointwise skinning seams and the D = 3 Broyden known-answer test of F2 / F1 on a subject whose skinning MLP
    has wide-range activations (hidden weight-norm gains x 8, config.widen_skinning_): what the f16-pair engine of loop C has
    to survive on weights it has not been tuned on."""
    torch.set_num_threads(os.cpu_count())
    scene = synthetic.SyntheticScene(seed=0)
    model, cfg = build_reference_model("zju377_mono")
    my_config.widen_skinning_(model, 8.0)
    inputs = scene.make_inputs(64, 64, frame_idx=0)
    ft = frame_tensors(model, inputs)
    loc, sc, vol = ft["loc"], ft["sc_factor"], ft["vol_feat"]
    cmin, cmax, center = inputs["coord_min"], inputs["coord_max"], inputs["center"]
    bones = inputs["bone_transforms"]
    skin = model.skinning_model
    g = torch.Generator().manual_seed(1234)
    P = 1024
    x_norm = torch.rand(1, P, 3, generator=g) * 1.6 - 0.8
    x_hat = RFU.unnormalize_canonical_points(x_norm, cmin, cmax, center)
    with torch.no_grad():
        dlog = skin.decode_w(x_norm, c=torch.empty(1, 0), forward=True)
        w = RFU.query_weights(x_hat, loc, sc, cmin, cmax, center, skin, vol)
        xbar, T = RFU.forward_skinning(x_hat, loc, sc, cmin, cmax, center, skin, vol, bones)
        Pb = 256
        xh_true = x_hat[:, :Pb]
        tgt, _ = RFU.forward_skinning(xh_true, loc, sc, cmin, cmax, center, skin, vol, bones)
        x0 = xh_true + torch.randn(1, Pb, 3, generator=g) * 0.01
        x0[:, :8] += 0.5

        def g3(x, mask=None):
            xx = x.reshape(1, Pb, 3)
            xb, Tt = RFU.forward_skinning(xx, loc, sc, cmin, cmax, center, skin, vol, bones, mask=mask)
            err = (xb - tgt).flatten(0, 1)[mask].unsqueeze(-1)
            return err, Tt.flatten(0, 1)[mask]

        w0 = RFU.query_weights(x0, loc, sc, cmin, cmax, center, skin, vol)
        T0 = torch.einsum("bpn,bnij->bpij", w0, bones)
        Jinv0 = T0[:, :, :3, :3].inverse()
        T_init = torch.eye(4).expand(Pb, 4, 4).clone() * 7.0
        r3 = ref_broyden(g3, x0.reshape(Pb, 3, 1), T_init, Jinv0.flatten(0, 1))
        # the largest hidden activation the reference's network produced on these points (for the record)
        h = skin.skinning_decoder_fwd
        act = x_norm[0]
        amax = []
        for k in range(4):
            lin = getattr(h, "lin%d" % k)
            act = torch.nn.functional.softplus(lin(act), beta=100)
            amax.append(float(act.abs().max()))
    save("f17_wide_skinning.npz", scale=8.0, x_norm=x_norm[0], x_hat=x_hat[0], deformer_logits=dlog[0], weights=w[0], x_bar=xbar[0],
         T=T[0], tgt=tgt[0], x0=x0[0], T0=T_init, result=r3["result"][:, :, 0], transforms=r3["transforms"], diff=r3["diff"],
         valid=r3["valid_ids"], hidden_absmax=np.array(amax, np.float32))
    print("hidden |activation| max per layer:", amax, "converged", int(r3["valid_ids"].sum()), "of", Pb)


F7_FULL = (("zju377_mono", (256, 256), (32, 8, 8), 1), ("zju377_mono", (512, 512), (64, 16, 16), 0),   # BASELINE configs 1 and 2
           ("h36m", (128, 128), (128, 32, 32), 4))   # round 6: BASELINE config 5's shapes and sampling on a 128 x 128 frame (`f7full 2`)
F7_SET = (("zju377_mono", (64, 64), (64, 16, 16), 0), ("zju313", (64, 64), (64, 16, 16), 1), ("h36m", (48, 48), (32, 8, 8), 2),
          ("zju377_mono", (128, 128), (32, 8, 8), 5), ("h36m", (40, 40), (128, 32, 32), 3))


def make_f18():
    """F18: the canonical-mesh branch of the model entry (models/__init__.py:203-311) as the reference runs it, with the
    third-party pieces replaced by RECORDERS: what the reference itself computes there -- the posed mesh vertices (its own
    unnormalisation, forward_skinning and translation), the arguments it hands to cameras_from_opencv_projection /
    look_at_view_transform / FoVPerspectiveCameras / RasterizationSettings, which normals it takes (sign, frame), how it
    colours the three 512 x 512 maps from pix_to_face -- is pinned by value.  Stand-ins (documented behaviour, unpinned as
    before): skimage's marching cubes -> the build's own (so the mesh is the build's, injected at N = 48 instead of 256 to
    keep the fixture small: create_mesh_vertices_and_faces is still the reference's), Meshes.faces_normals_packed (unit
    right-hand normals), the rasteriser -> oracle/mesh_oracle.rasterize_np on the build's projections."""
    import skimage.measure as skm
    from im2mesh.utils import sdf_meshing
    from im2mesh.metaavatar_render import models as ref_models
    sys.path.insert(0, REPO)
    from arah_release_amd import meshing
    from oracle import mesh_oracle
    rec = {"raster": [], "lookat": [], "fov": [], "settings": []}
    N_INJECT = 48

    def fake_mc(volume, level=None, spacing=None, **kw):
        tri = meshing.marching_cubes(torch.from_numpy(np.ascontiguousarray(volume)).float(), level=float(level))   # (F,3,3) in [-1,1]^3
        rec["tri"] = tri.numpy().copy()
        verts = (tri.reshape(-1, 3).numpy().astype(np.float64) + 1.0)        # skimage returns index * spacing from the origin
        faces = np.arange(verts.shape[0]).reshape(-1, 3)
        return verts, faces, np.zeros_like(verts), np.zeros(verts.shape[0])

    real_create = sdf_meshing.create_mesh_vertices_and_faces

    def create_small(decoder, N=256, max_batch=64 ** 3, **kw):
        rec["asked_N"], rec["asked_max_batch"] = N, max_batch
        return real_create(decoder, N=N_INJECT, max_batch=max_batch, **kw)

    class Cameras:
        def __init__(self, kind, **kw):
            self.kind, self.kw = kind, kw

        def to(self, device):
            return self

    def cameras_from_opencv_projection(R, tvec, camera_matrix, image_size):
        rec["opencv"] = dict(R=R.clone(), tvec=tvec.clone(), K=camera_matrix.clone(), image_size=image_size.clone())
        return Cameras("opencv", R=R[0], t=tvec[0], K=camera_matrix[0])

    def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, **kw):
        rec["lookat"].append((float(dist), float(elev), float(azim)))
        return ("R", float(azim)), ("T", float(dist))

    def FoVPerspectiveCameras(device=None, R=None, T=None, **kw):
        rec["fov"].append(dict(R=R, T=T, extra=sorted(kw)))
        return Cameras("fov", azim=R[1], dist=T[1])

    class RasterizationSettings:
        def __init__(self, image_size=256, **kw):
            rec["settings"].append((image_size, sorted(kw)))
            self.image_size = image_size

    class Meshes:
        def __init__(self, verts, faces):
            self.verts, self.faces = verts, faces
            rec.setdefault("meshes", []).append((verts.clone(), faces.clone()))

        def faces_normals_packed(self):
            v = self.verts[0]
            f = self.faces[0].long()
            n = torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=1)
            return n / n.norm(dim=1, keepdim=True).clamp_min(1e-20)

    class Fragments:
        def __init__(self, p2f):
            self.pix_to_face = p2f

    class MeshRasterizer:
        def __init__(self, cameras=None, raster_settings=None):
            self.cam, self.rs = cameras, raster_settings

        def __call__(self, mesh):
            size = self.rs.image_size
            H, W = (size, size) if isinstance(size, int) else size
            tri = mesh.verts[0][mesh.faces[0].long()]                                   # (F,3,3)
            if self.cam.kind == "opencv":
                uvz = meshing.project_opencv(tri, self.cam.kw["R"], self.cam.kw["t"], self.cam.kw["K"])
                p2f = mesh_oracle.rasterize_np(uvz.numpy(), H, W)
            else:
                uvz = meshing.project_lookat(tri, self.cam.kw["azim"], H, dist=self.cam.kw["dist"])
                p2f = mesh_oracle.rasterize_np(uvz.numpy(), H, W, z_near=1.0)
            rec["raster"].append(dict(uvz=uvz.numpy().copy(), p2f=p2f.copy(), H=H, W=W))
            return Fragments(torch.from_numpy(p2f).reshape(1, H, W, 1))

    mods = {"pytorch3d.utils": dict(cameras_from_opencv_projection=cameras_from_opencv_projection),
            "pytorch3d.structures": dict(Meshes=Meshes),
            "pytorch3d.renderer": dict(look_at_view_transform=look_at_view_transform, PerspectiveCameras=Cameras,
                                       FoVPerspectiveCameras=FoVPerspectiveCameras, RasterizationSettings=RasterizationSettings,
                                       MeshRenderer=Cameras, MeshRasterizer=MeshRasterizer, TexturesVertex=Cameras)}
    saved = {}
    for m, attrs in mods.items():
        for k, v in attrs.items():
            saved[(m, k)] = sys.modules[m].__dict__.get(k)
            setattr(sys.modules[m], k, v)
    old_cuda, old_mc = torch.Tensor.cuda, getattr(skm, "marching_cubes_lewiner", None)
    torch.Tensor.cuda = lambda self, *a, **k: self
    skm.marching_cubes_lewiner = fake_mc
    sdf_meshing.create_mesh_vertices_and_faces = create_small
    try:
        model, cfg = build_reference_model("zju377_mono", 64, 16, 16)
        scene = synthetic.SyntheticScene(0)
        fidx, H, W, max_rays = 5, 512, 512, 256          # the branch rasterises at 512 x 512 with the frame's intrinsics
        inputs = scene.make_inputs(H, W, frame_idx=fidx, max_rays=max_rays)
        with torch.no_grad():
            out = model(inputs, gen_cano_mesh=True, eval=True)
    finally:
        torch.Tensor.cuda = old_cuda
        sdf_meshing.create_mesh_vertices_and_faces = real_create
        if old_mc is not None:
            skm.marching_cubes_lewiner = old_mc
        for (m, k), v in saved.items():
            if v is None:
                sys.modules[m].__dict__.pop(k, None)
            else:
                setattr(sys.modules[m], k, v)
    assert rec["asked_N"] == 256 and rec["asked_max_batch"] == 64 ** 3            # models/__init__.py:205-206
    assert rec["lookat"] == [(2.0, 0.0, 0.0), (2.0, 0.0, 180.0)] and all(f["extra"] == [] for f in rec["fov"])
    assert rec["settings"] == [((512, 512), []), (512, [])] and len(rec["raster"]) == 3 and len(rec["meshes"]) == 3
    posed, faces0 = rec["meshes"][0]
    cano, _ = rec["meshes"][1]
    assert torch.equal(rec["meshes"][1][0], rec["meshes"][2][0])
    tri = rec["tri"]
    assert np.allclose(cano[0].numpy().reshape(-1, 3, 3), tri, atol=1e-6)           # the mesh the reference rasterises IS the injected one
    save("f18_cano_mesh_branch.npz", frame_idx=fidx, H=H, W=W, max_rays=max_rays, n_inject=N_INJECT, tri=tri.astype(np.float32),
         posed_verts=posed[0], opencv_R=rec["opencv"]["R"], opencv_t=rec["opencv"]["tvec"], opencv_K=rec["opencv"]["K"],
         opencv_image_size=rec["opencv"]["image_size"], cam_rot=inputs["cam_rot"], cam_trans=inputs["cam_trans"],
         intrinsics=inputs["intrinsics"],
         p2f_posed=rec["raster"][0]["p2f"].astype(np.int32), p2f_front=rec["raster"][1]["p2f"].astype(np.int32),
         p2f_back=rec["raster"][2]["p2f"].astype(np.int32),
         output_normal=out["output_normal"].numpy().astype(np.float32), normal_cano_front=out["normal_cano_front"].numpy().astype(np.float32),
         normal_cano_back=out["normal_cano_back"].numpy().astype(np.float32))


def make_f19():
    """F19: the training-time depth sampler alone (ray_tracing.py:298-350: perturb_z_vals + ray_sampler's z_vals) -- the
    reference's BodyRayTracing.ray_sampler called with its canonicalisation stubbed out (generate_point_samples_opt is F5's
    business), its three torch.rand draws recorded.  Rays with and without a surface hit, a surface closer to the near bound
    than the surface range (the 1e-5 floor of RT:346), two sampling configurations."""
    from im2mesh.metaavatar_render.renderer.ray_tracing import BodyRayTracing
    out = {}
    for tag, (S, n_near, n_far) in (("s64", (64, 16, 16)), ("s32", (32, 8, 8))):
        g = torch.Generator().manual_seed(77 + S)
        N = 257
        near = 2.0 + torch.rand(1, N, generator=g)
        far = near + 0.5 + torch.rand(1, N, generator=g)
        conv = torch.rand(1, N, generator=g) > 0.4
        surf = near + (far - near) * torch.rand(1, N, generator=g)
        surf[0, :8] = near[0, :8] + 0.01                      # closer to the near bound than the 5 cm surface range
        start = torch.where(conv, surf, near)
        tracer = BodyRayTracing(n_steps=S, near_surface_vol_samples=n_near, far_surface_vol_samples=n_far)
        tracer.generate_point_samples_opt = lambda *a, **k: (None, None, None)
        draws, orig = [], torch.rand

        def recording_rand(*a, **k):
            t = orig(*a, **k)
            draws.append(t.detach().clone())
            return t

        torch.manual_seed(4321 + S)
        torch.rand = recording_rand
        try:
            _, _, _, z = tracer.ray_sampler(None, None, None, conv, None, torch.stack([start, far], dim=-1),
                                            torch.stack([near, far], dim=-1), torch.ones_like(conv), *([None] * 11), eval_mode=False)
        finally:
            torch.rand = orig
        assert [tuple(d.shape) for d in draws] == [(1, N, S), (1, N, n_near + 1), (1, N, n_far)], [tuple(d.shape) for d in draws]
        out.update({tag + ".conv": conv[0], tag + ".start": start[0], tag + ".end": far[0], tag + ".near": near[0],
                    tag + ".rand_steps": draws[0][0], tag + ".rand_near": draws[1][0], tag + ".rand_far": draws[2][0],
                    tag + ".z": z[0], tag + ".cfg": torch.tensor([S, n_near, n_far])})
    save("f19_jitter_depths.npz", **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "f19":
        return make_f19()
    if len(sys.argv) > 1 and sys.argv[1] == "f18":
        return make_f18()
    if len(sys.argv) > 1 and sys.argv[1] == "f7full":
        return make_f7(F7_FULL[int(sys.argv[2]):int(sys.argv[2]) + 1] if len(sys.argv) > 2 else F7_FULL, full_frames=True)
    if len(sys.argv) > 1 and sys.argv[1] == "f7c5":
        return make_f7(F7_SET[-1:])
    if len(sys.argv) > 1 and sys.argv[1] == "f17":
        return make_f17()
    if len(sys.argv) > 1 and sys.argv[1] == "f16":
        return make_f16()
    if len(sys.argv) > 1 and sys.argv[1] == "f15":
        return make_f15()
    if len(sys.argv) > 1 and sys.argv[1] == "f14":
        return make_f14()
    if len(sys.argv) > 1 and sys.argv[1] == "f13":
        return make_f13()
    if len(sys.argv) > 1 and sys.argv[1] == "f12":
        return make_f12()
    if len(sys.argv) > 1 and sys.argv[1] == "f11":
        return make_f11()
    if len(sys.argv) > 1 and sys.argv[1] == "f10":
        return make_f10()
    if len(sys.argv) > 1 and sys.argv[1] == "f9":
        return make_f9()
    if len(sys.argv) > 1 and sys.argv[1] == "f1d4":
        return make_f1_d4()
    if len(sys.argv) > 1 and sys.argv[1] == "f8":
        return make_f8()
    torch.set_num_threads(os.cpu_count())
    scene = synthetic.SyntheticScene(seed=0)
    model, cfg = build_reference_model("zju377_mono")
    inputs = scene.make_inputs(64, 64, frame_idx=0)
    ft = frame_tensors(model, inputs)
    sdf_network, loc, sc, vol = ft["sdf_network"], ft["loc"], ft["sc_factor"], ft["vol_feat"]
    cmin, cmax, center = inputs["coord_min"], inputs["coord_max"], inputs["center"]
    bones, trans = inputs["bone_transforms"], inputs["trans"]
    skin = model.skinning_model
    g = torch.Generator().manual_seed(123)

    # ------------------------------------------------------------------ F2 pointwise pieces
    P = 1024
    logits = torch.randn(1, P, 25, generator=g) * 3
    x_norm = torch.rand(1, P, 3, generator=g) * 1.6 - 0.8
    x_hat = RFU.unnormalize_canonical_points(x_norm, cmin, cmax, center)
    with torch.no_grad():
        hs = ref_hsoftmax(logits)
        xn_back = RFU.normalize_canonical_points(x_hat, cmin, cmax, center)
        dlog = skin.decode_w(x_norm, c=torch.empty(1, 0), forward=True)
        w = RFU.query_weights(x_hat, loc, sc, cmin, cmax, center, skin, vol)
        xbar, T = RFU.forward_skinning(x_hat, loc, sc, cmin, cmax, center, skin, vol, bones)
    jac = RFU.forward_skinning_jac(x_hat.clone(), loc, sc, cmin, cmax, center, skin, vol, bones)
    save("f2_pointwise.npz", logits=logits[0], hsoftmax=hs[0], x_norm=x_norm[0], x_hat=x_hat[0],
         x_norm_back=xn_back[0], deformer_logits=dlog[0], weights=w[0], x_bar=xbar[0], T=T[0], jac=jac[0],
         coord_min=cmin.reshape(-1), coord_max=cmax.reshape(-1), center=center.reshape(-1), bones=bones[0])

    # ------------------------------------------------------------------ F3 SDF MLP
    with torch.enable_grad():
        xg = x_norm.clone().requires_grad_(True)
        feat = sdf_network[:-1](xg)
        sdf = sdf_network[-1](feat)
        grad = diff_operators.gradient(sdf, xg, create_graph=False)
    save("f3_sdf.npz", x_norm=x_norm[0], sdf=sdf[0, :, 0], feat=feat[0], grad=grad[0])

    # ------------------------------------------------------------------ F1 Broyden KATs
    # D=3: g = LBS(x) - target, as in search_canonical_corr (root_finding_utils.py:267-303)
    Pb = 256
    xh_true = x_hat[:, :Pb]
    with torch.no_grad():
        tgt, _ = RFU.forward_skinning(xh_true, loc, sc, cmin, cmax, center, skin, vol, bones)
        x0 = xh_true + torch.randn(1, Pb, 3, generator=g) * 0.02
        x0[:, :8] += 0.5   # a few hopeless starts exercise the divergence / best-iterate paths
        w0 = RFU.query_weights(x0, loc, sc, cmin, cmax, center, skin, vol)
        T0 = torch.einsum("bpn,bnij->bpij", w0, bones)
        Jinv0 = T0[:, :, :3, :3].inverse()

        def g3(x, mask=None):
            xx = x.reshape(1, Pb, 3)
            xb, Tt = RFU.forward_skinning(xx, loc, sc, cmin, cmax, center, skin, vol, bones, mask=mask)
            err = (xb - tgt).flatten(0, 1)[mask].unsqueeze(-1)
            return err, Tt.flatten(0, 1)[mask]

        T_init = torch.eye(4).expand(Pb, 4, 4).clone() * 7.0   # sentinel: returned when the start is already best
        r3 = ref_broyden(g3, x0.reshape(Pb, 3, 1), T_init, Jinv0.flatten(0, 1))
    save("f1_broyden3.npz", tgt=tgt[0], x0=x0[0], T0=T_init, Jinv0=Jinv0[0], result=r3["result"][:, :, 0],
         transforms=r3["transforms"], diff=r3["diff"], valid=r3["valid_ids"])

    for tag, (S, near, far) in (("s64", (64, 16, 16)), ("s32", (32, 8, 8))):
        for name in ("zju377_mono", "h36m"):
            if tag == "s32" and name == "h36m":
                continue
            model, cfg = build_reference_model(name, S, near, far)
            inputs = scene.make_inputs(96, 96, frame_idx=3, max_rays=1024)
            ft = frame_tensors(model, inputs)
            tracer = model.idhr_network.ray_tracer
            with torch.no_grad():
                tr = tracer(ft["sdf_network"], model.skinning_model, cam_loc=inputs["cam_loc"],
                            ray_directions=inputs["ray_dirs"],
                            body_bounds_intersections=inputs["body_bounds_intersections"], loc=ft["loc"],
                            sc_factor=ft["sc_factor"], smpl_verts=inputs["smpl_verts"],
                            smpl_verts_cano=inputs["minimal_shape"], skinning_weights=inputs["skinning_weights"],
                            vol_feat=ft["vol_feat"], bone_transforms=inputs["bone_transforms"],
                            trans=inputs["trans"], coord_min=inputs["coord_min"], coord_max=inputs["coord_max"],
                            center=inputs["center"], eval_mode=True)
            xn, nbm, dists, spts, sz, sT, sm = tr
            # bottom row of every valid blended transform is (0,0,0,sum w ~ 1): store the 3x4 part only
            bot = sT[0][sm[0]][:, 3]
            assert float((bot - torch.tensor([0., 0., 0., 1.])).abs().max()) < 1e-5
            if name == "zju377_mono":
                save("f5_tracer_%s.npz" % tag, frame_idx=3, H=96, W=96, max_rays=1024, n_steps=S, n_near=near,
                     n_far=far, points_hat_norm=xn[0], network_body_mask=nbm[0], dists=dists[0], sampler_pts=spts[0],
                     sampler_dists=sz[0], sampler_transforms34=sT[0][:, :, :3, :].reshape(sT.shape[1], S, 12),
                     sampler_converge_mask=sm[0])
            # ---------------------------------------------------------- F6 shading on the tracer output
            vol_mask = sm[0].any(-1)
            pose_cond = dict(inputs["pose_cond"])
            pose_cond["latent_code"] = model.latent(pose_cond["latent_code_idx"])
            idhr = model.idhr_network
            rgb, acc = idhr.get_rbg_value_vol_sdf(ft["sdf_network"], spts[0][vol_mask], sz[0][vol_mask],
                                                  sT[0][vol_mask], sm[0][vol_mask], inputs["ray_dirs"][0][vol_mask],
                                                  inputs["ray_dirs"][0][vol_mask].clone(), pose_cond, ft["loc"][:1],
                                                  ft["sc_factor"][:1], ft["vol_feat"], inputs["bone_transforms"][:1],
                                                  inputs["coord_min"][:1], inputs["coord_max"][:1],
                                                  inputs["center"][:1], point_batch_size=1000000)
            # inputs of this call are the f5_tracer_<tag> arrays (the tracer does not depend on the colour net)
            save("f6_shade_%s_%s.npz" % (name, tag), vol_mask=vol_mask, rgb=rgb.detach(), acc=acc.detach(),
                 frame_idx=3, H=96, W=96, max_rays=1024, n_steps=S, n_near=near, n_far=far)

    # ------------------------------------------------------------------ F4 colour MLP (both modes)
    for name in ("zju377_mono", "zju313"):
        model, cfg = build_reference_model(name)
        inputs = scene.make_inputs(64, 64, frame_idx=0)
        pose_cond = dict(inputs["pose_cond"])
        pose_cond["latent_code"] = model.latent(pose_cond["latent_code_idx"])
        pts = torch.rand(P, 3, generator=g) * 2 - 1
        nrm = torch.randn(P, 3, generator=g)
        view = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
        feat = torch.rand(P, 256, generator=g) * 2 - 1
        with torch.no_grad():
            rgb = model.color_decoder(pts, nrm, view, feat, pose_cond)
        save("f4_color_%s.npz" % name, points=pts, normals=nrm, view=view, feat=feat, rgb=rgb)

    # ------------------------------------------------------------------ F7 whole forward
    make_f7(F7_SET)


if __name__ == "__main__":
    main()
