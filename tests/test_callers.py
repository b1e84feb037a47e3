"""Callers and data formats either side of the hot path (VERDICT "missing" #3): SMPL linear blend
skinning, Vitruvian-pose transforms, ray / box helpers and LightningModel.compose_inputs against fixture F9 (generated
by the reference's own human_body_prior.lbs, get_transforms_02v / get_02v_bone_transforms, get_near_far,
get_camera_rays / get_camera_location and LightningModel.compose_inputs in both branches); the on-disk formats; the
train_smpl / train_cameras construction; the pre-trained initialisation of get_model."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import golden

T = lambda a: torch.from_numpy(np.asarray(a, np.float32))


@pytest.fixture(scope="module")
def body(scene):
    from arah_release_amd import smpl
    return smpl.BodyModel.synthetic(scene)


def test_lbs_against_reference(body):
    from arah_release_amd import smpl
    g = golden("f9_callers.npz")
    verts, J_posed, J, A, _, v_posed = smpl.lbs(T(g["lbs_betas"]), T(g["lbs_pose"]), T(body.v_template)[None],
                                                T(body.shapedirs), T(body.posedirs), T(body.J_regressor),
                                                torch.from_numpy(body.kintree_table[0].astype(np.int64)),
                                                T(body.lbs_weights))
    for got, key in ((verts, "lbs_verts"), (J_posed, "lbs_J_posed"), (J, "lbs_J"), (A, "lbs_A"), (v_posed, "lbs_v_posed")):
        np.testing.assert_allclose(got[0].numpy(), g[key], rtol=1e-5, atol=1e-6)


def test_vitruvian_transforms_against_reference():
    from arah_release_amd import smpl
    g = golden("f9_callers.npz")
    got = smpl.get_transforms_02v(T(g["lbs_J"])).numpy()
    np.testing.assert_allclose(got, g["v02_torch"], rtol=1e-6, atol=1e-6)      # lightning_model.py:37-99
    np.testing.assert_allclose(got, g["v02_numpy"], rtol=1e-5, atol=1e-6)      # the dataset's numpy twin


def test_rays_and_box_against_reference():
    from arah_release_amd import data
    g = golden("f9_callers.npz")
    near, far, ok = data.near_far(T(g["nf_bounds"]), T(g["nf_ray_o"]).expand(512, 3), T(g["nf_ray_d"]))
    np.testing.assert_array_equal(ok.numpy(), g["nf_ok"])
    np.testing.assert_allclose(near.numpy(), g["nf_near"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(far.numpy(), g["nf_far"], rtol=1e-5, atol=1e-5)
    R, uv = T(g["cam_R"]), T(g["cam_uv"])
    rays = uv @ R
    rays = rays / (rays.norm(dim=-1, keepdim=True) + 1e-12)
    np.testing.assert_allclose(rays.numpy(), g["cam_rays"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose((-R.t() @ T([0.3, -0.1, 2.0])).numpy(), g["cam_loc"], rtol=1e-6, atol=1e-6)


def _item(g, body):
    from arah_release_amd import data
    md = {k[3:]: g[k] for k in g.files if k.startswith("md.")}
    cam = {k[4:]: g[k] for k in g.files if k.startswith("cam.")}
    return data.frame_item(md, cam, body, 64, 64, frame_idx=5, data_idx=1)


_KEEP = ("ray_dirs", "cam_loc", "pose", "bone_transforms", "trans", "coord_min", "coord_max", "center", "Jtrs", "rots",
         "smpl_verts", "minimal_shape", "cam_rot", "cam_trans")


def test_compose_inputs_dataset_branch_against_reference(body):
    """Dataset-provided SMPL, eval: lightning_model.py:463-634 with train_smpl / train_cameras off."""
    from arah_release_amd import config
    g = golden("f9_callers.npz")
    cfg = config.builtin_config("zju313")
    lm = config.get_model(cfg, mode="test", n_data_points=4)
    lm.model.frames = []
    item = _item(g, body)
    inp = lm.compose_inputs(item, eval=True)
    for k in _KEEP:
        np.testing.assert_allclose(inp[k].numpy(), g["ciA." + k], rtol=1e-5, atol=1e-5, err_msg=k)
    np.testing.assert_allclose(inp["pose_cond"]["rots_full"].numpy(), g["ciA.rots_full"], rtol=1e-5, atol=1e-6)
    assert int(inp["pose_cond"]["latent_code_idx"]) == int(np.ravel(g["ciA.latent_code_idx"])[0]) == 3     # unseen frame: last code
    assert int(inp["geo_latent_code_idx"]) == int(np.ravel(g["ciA.geo_latent_code_idx"])[0])
    assert "image_mask" in inp and "ray_dirs_cam" in inp and "rgb_values" not in inp


def test_compose_inputs_optimised_smpl_and_cameras_against_reference(body):
    """train_smpl + train_cameras, training: the SMPL parameters and the camera extrinsics come from the model's own
    nn.Parameters (forward_smpl -> LBS, Vitruvian transforms, re-normalisation; quaternion -> rays)."""
    from arah_release_amd import config
    g = golden("f9_callers.npz")

    class DS:
        cam_names = ["0"]
        cameras = {"0": {"R": np.eye(3), "T": np.zeros(3)}}
        data = []

    cfg = config.builtin_config("zju313")
    cfg["model"].update(train_smpl=False, train_cameras=False)
    lm = config.get_model(cfg, mode="test", n_data_points=4)
    m = lm.model
    # the constructor path is exercised in test_train_smpl_construction; here the fixture's parameter values go in
    from arah_release_amd import renderer
    kw = dict(frames=[4, 5], betas=g["ciB.betas"], body_model=body, cam_rots=g["ciB.cam_rots"], cam_trans=g["ciB.cam_trans"],
              n_data_points=4)
    for key in ("root_orient", "pose_body", "pose_hand", "trans"):
        kw[key] = [g["ciB.%s_%d" % (key, fr)] for fr in (4, 5)]
    m2 = renderer.MetaAvatarRender(m.sdf_decoder, m.skinning_model, m.color_decoder, m.deviation_decoder, train_cameras=True,
                                   train_smpl=True, train_latent_code=True, train_geo_latent_code=True, **kw)
    lm.model = m2
    item = _item(g, body)
    item["inputs.novel_seq"] = None
    item["inputs"] = T(g["ciB.rgb_values"])
    inp = lm.compose_inputs(item, eval=False)
    for k in _KEEP:
        np.testing.assert_allclose(inp[k].detach().numpy(), g["ciB." + k], rtol=2e-5, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(inp["pose_cond"]["rots_full"].detach().numpy(), g["ciB.rots_full"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(inp["pose_cond"]["Jtrs_posed"].detach().numpy(), g["ciB.Jtrs_posed"], rtol=1e-5, atol=1e-5)
    assert int(inp["pose_cond"]["latent_code_idx"]) == int(np.ravel(g["ciB.latent_code_idx"])[0]) == 1     # seen frame: its data index
    # the inputs depend on the optimised parameters
    inp["smpl_verts"].sum().backward()
    assert m2.body_poses["pose_body_5"].grad is not None and m2.betas.grad is not None
    assert m2.body_poses["pose_body_4"].grad is None


def test_bound_mask_is_the_projected_box(body):
    """Pixels of the projected box: every pixel whose ray hits the box is inside the mask (the mask is what the reference
    then filters with near < far), and the mask is not much larger than that set."""
    from arah_release_amd import data
    g = golden("f9_callers.npz")
    item = _item(g, body)
    mask = item["inputs.image_mask"][0]
    H, W = mask.shape
    assert 0.05 < float(mask.float().mean()) < 0.9
    n = item["inputs.ray_dirs"].shape[1]
    assert int(mask.sum()) == n
    nf = item["inputs.body_bounds_intersections"][0]
    assert bool((nf[:, 0] < nf[:, 1]).all())
    d = item["inputs.ray_dirs"][0]
    np.testing.assert_allclose(d.norm(dim=-1).numpy(), 1.0, atol=1e-5)


def test_readers_on_files_written_by_the_reference():
    """Fixture F16: models/000000.npz and cam_params.json as the reference's OWN preprocessing script writes them
    (preprocess_datasets/preprocess_ZJU-MoCap.py:150-164, run on synthetic inputs by tests/golden/make_golden.py f16).
    The readers must take them as they are: key names, shapes, dtypes, the metre translations, the JSON nesting."""
    from arah_release_amd import data
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f16_preprocessed_CoreView_377")
    src = np.load(os.path.join(root, "inputs.npz"))
    frames, files = data.list_sequence(root)
    assert frames == [0] and files[0].endswith("models/000000.npz")
    m = data.load_model_npz(files[0])
    assert set(m) == set(data.MODEL_KEYS)
    for k, shp in data.MODEL_SHAPES.items():
        assert m[k].shape == shp and m[k].dtype == np.float32, k
    assert m["betas"].shape == (1, 10)
    np.testing.assert_array_equal(m["minimal_shape"], src["verts"])          # what the (stand-in) body model returned
    np.testing.assert_array_equal(m["Jtr_posed"], src["Jtr"])
    np.testing.assert_array_equal(m["bone_transforms"], src["bone_transforms"])
    assert m["pose_body"].shape == (63,) and m["pose_hand"].shape == (6,) and m["root_orient"].shape == (3,)
    cams = data.load_cam_params(os.path.join(root, "cam_params.json"))
    assert cams["all_cam_names"] == [str(c) for c in range(1, 24)]            # preprocess_ZJU-MoCap.py:46-49
    for c, name in enumerate(cams["all_cam_names"]):
        cam = cams[name]
        np.testing.assert_allclose(cam["K"], src["K"][c], rtol=1e-6)
        np.testing.assert_allclose(cam["R"], src["R"][c], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(cam["T"], src["T"][c].reshape(3) / 1000.0, rtol=1e-6)   # millimetres -> metres (:70-71)
        np.testing.assert_allclose(cam["D"], src["D"][c].reshape(-1), rtol=1e-6, atol=1e-9)


def test_on_disk_formats_round_trip(tmp_path, scene, body):
    """cam_params.json / models/*.npz as preprocess_ZJU-MoCap.py:150-158 writes them."""
    from arah_release_amd import data
    sub = tmp_path / "CoreView_000"
    (sub / "models").mkdir(parents=True)
    rng = np.random.RandomState(0)
    for f in range(5):
        np.savez(sub / "models" / ("%06d.npz" % f), minimal_shape=scene.verts_cano.astype(np.float16 if f == 0 else np.float32),
                 betas=np.zeros((1, 10), np.float32), Jtr_posed=scene.joints, bone_transforms=np.tile(np.eye(4, dtype=np.float32), (24, 1, 1)),
                 trans=np.array([0.1, 0, 3.0], np.float32), root_orient=rng.randn(3).astype(np.float32),
                 pose_body=rng.randn(63).astype(np.float32), pose_hand=np.zeros(6, np.float32))
    cams = {"all_cam_names": ["1", "7"], "1": {"K": np.eye(3).tolist(), "D": [0] * 5, "R": np.eye(3).tolist(), "T": [[0], [0], [1.0]]},
            "7": {"K": np.eye(3).tolist(), "D": [0] * 5, "R": np.eye(3).tolist(), "T": [[0], [0], [2.0]]}}
    (sub / "cam_params.json").write_text(json.dumps(cams))
    cp = data.load_cam_params(str(sub / "cam_params.json"))
    assert cp["all_cam_names"] == ["1", "7"] and cp["7"]["T"].shape == (3,) and cp["7"]["T"][2] == 2.0
    frames, files = data.list_sequence(str(sub), start_frame=1, end_frame=5, sampling_rate=2)
    assert frames == [1, 3] and [os.path.basename(f) for f in files] == ["000001.npz", "000003.npz"]
    md = data.load_model_npz(str(sub / "models" / "000000.npz"))
    assert md["minimal_shape"].dtype == np.float32 and md["pose_body"].shape == (63,)
    np.savez(sub / "models" / "bad.npz", minimal_shape=np.zeros((10, 3)))
    with pytest.raises(ValueError):
        data.load_model_npz(str(sub / "models" / "bad.npz"))
    bad = dict(cams)
    del bad["7"]
    (sub / "bad.json").write_text(json.dumps(bad))
    with pytest.raises(ValueError):
        data.load_cam_params(str(sub / "bad.json"))


def test_train_smpl_construction_from_a_dataset(tmp_path, scene, body):
    """get_model(cfg, dataset=..., mode='train') with train_smpl / train_cameras (config.py:166-224, models/__init__.py:
    81-123): one ParameterDict entry per frame of the first camera, exact zeros nudged, quaternions from the cameras."""
    from arah_release_amd import config
    from scipy.spatial.transform import Rotation
    sub = tmp_path / "s"
    (sub / "models").mkdir(parents=True)
    files = []
    for f in range(3):
        p = sub / "models" / ("%06d.npz" % f)
        np.savez(p, minimal_shape=scene.verts_cano, betas=np.full((1, 10), 0.1, np.float32), Jtr_posed=scene.joints,
                 bone_transforms=np.tile(np.eye(4, dtype=np.float32), (24, 1, 1)), trans=np.array([0, 0, 3.0 + f], np.float32),
                 root_orient=np.zeros(3, np.float32), pose_body=np.full(63, 0.01 * f, np.float32), pose_hand=np.zeros(6, np.float32))
        files.append(str(p))

    class DS:
        cam_names = ["a", "b"]
        cameras = {"a": {"R": np.eye(3), "T": [[0.0], [0.0], [0.0]]},
                   "b": {"R": Rotation.from_rotvec([0, 0.3, 0]).as_matrix(), "T": [[1.0], [0.0], [0.0]]}}
        data = [{"cam_idx": c, "frame_idx": 10 + f, "data_idx": f, "model_file": files[f], "gender": "neutral"}
                for c in range(2) for f in range(3)]

    cfg = config.builtin_config("zju313")
    cfg["model"].update(train_smpl=True, train_cameras=True)
    lm = config.get_model(cfg, dataset=DS, mode="train", body_model=body)
    m = lm.model
    assert m.frames == [10, 11, 12] and m.latent.num_embeddings == 3
    assert set(m.body_poses.keys()) == {"%s_%d" % (k, f) for k in ("root_orient", "pose_body", "pose_hand", "trans") for f in (10, 11, 12)}
    assert float(m.body_poses["root_orient_10"].detach().abs().max()) == pytest.approx(1e-8)       # zero rotation got its nudge
    assert float(m.body_poses["trans_12"].detach()[2]) == 5.0 and tuple(m.betas.shape) == (1, 10)
    np.testing.assert_allclose(m.cam_rots[1].detach().numpy(), Rotation.from_rotvec([0, 0.3, 0]).as_quat(), atol=1e-6)
    assert len(list(m.smpl_parameters())) == 13 and len(list(m.camera_parameters())) == 2
    n_all = len(list(m.parameters()))
    assert len(list(m.network_parameters())) == n_all - 3           # cam_rots, cam_trans, betas (as in the reference)
    assert len(lm.configure_optimizers().param_groups) == 8
    # without the files and without an injected body model the reference's file lookup fails loudly
    with pytest.raises(FileNotFoundError):
        config.get_model(cfg, dataset=DS, mode="train")
    # val / test construction takes none of this
    assert not config.get_model(cfg, mode="test", n_data_points=3).model.train_smpl


def test_pretrained_initialisation_is_loaded_or_refused(tmp_path):
    """metaavatar_render/config.py:18-84: mode 'train' loads cfg['model']['geometry_net'] / ['skinning_net2']."""
    from arah_release_amd import config
    cfg = config.builtin_config("zju377_mono")
    ref = config.get_model(cfg, mode="test", n_data_points=2).model
    geo = {"module.decoder." + k: torch.full_like(v, 0.25) for k, v in ref.sdf_decoder.state_dict().items()}
    geo["module.encoder.something"] = torch.zeros(1)
    skin = {"skinning_decoder_fwd." + k: torch.full_like(v, 0.5) for k, v in ref.skinning_model.skinning_decoder_fwd.state_dict().items()}
    torch.save({"model": geo}, tmp_path / "geo.pt")
    torch.save({"model": skin}, tmp_path / "skin.pt")
    cfg["model"].update(geometry_net=str(tmp_path / "geo.pt"), skinning_net2=str(tmp_path / "skin.pt"))
    m = config.get_model(cfg, mode="train", n_data_points=2).model
    assert all(bool((v == 0.25).all()) for v in m.sdf_decoder.state_dict().values())
    assert all(bool((v == 0.5).all()) for v in m.skinning_model.skinning_decoder_fwd.state_dict().values())
    m_test = config.get_model(cfg, mode="test", n_data_points=2).model               # val / test: no initialisation
    assert not all(bool((v == 0.25).all()) for v in m_test.sdf_decoder.state_dict().values())
    cfg["model"]["geometry_net"] = str(tmp_path / "missing.pt")
    with pytest.raises(FileNotFoundError):
        config.get_model(cfg, mode="train", n_data_points=2)
    torch.save({"model": {"decoder.unrelated": torch.zeros(1)}}, tmp_path / "empty.pt")
    cfg["model"]["geometry_net"] = str(tmp_path / "empty.pt")
    with pytest.raises(ValueError):
        config.get_model(cfg, mode="train", n_data_points=2)


@pytest.mark.gpu
def test_dataset_item_renders_on_the_device(body):
    """frame_item -> compose_inputs -> MetaAvatarRender.forward on cuda:0: the callers' side feeds the hot path without a
    host copy of the frame (the item moves to the device once; everything after is device tensors)."""
    from arah_release_amd import config
    g = golden("f9_callers.npz")
    dev = torch.device("cuda:0")
    cfg = config.builtin_config("zju313")
    lm = config.get_model(cfg, mode="test", n_data_points=4).to(dev).eval()
    lm.model.frames = []
    item = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in _item(g, body).items()}
    inp = lm.compose_inputs(item, eval=True)
    assert inp["ray_dirs"].is_cuda and inp["bone_transforms"].is_cuda
    with torch.no_grad():
        out = lm.model(inp, gen_cano_mesh=False, eval=True)
    n = inp["ray_dirs"].shape[1]
    assert tuple(out["rgb_values"].shape) == (1, n, 3) and bool(torch.isfinite(out["rgb_values"]).all())
    assert 0 < int(out["network_body_mask"].sum()) <= n


def test_training_rays_sampling_properties():
    """zju_mocap.py:330-400 on tensors: counts, every kept ray hits the box, foreground pixels come from the eroded body
    mask with their colours, background pixels from the projected box outside it and black, rays through the pixels."""
    from arah_release_amd import data
    H = W = 256
    K = torch.tensor([[300.0, 0, 128], [0, 300.0, 128], [0, 0, 1]])
    R, T = torch.eye(3), torch.tensor([0.0, 0.0, 3.0])
    bounds = torch.tensor([[-0.45, -0.95, -0.25], [0.45, 0.95, 0.25]])
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    body = ((xx - 128).abs() < 26) & ((yy - 128).abs() < 70)
    rim = ((xx - 128).abs() < 29) & ((yy - 128).abs() < 73) & ~body
    me = torch.zeros(H, W, dtype=torch.int64)
    me[body], me[rim] = 1, 100
    mask = (body | rim).long()
    img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(0))
    out = data.training_rays(img, mask, me, bounds, K, R, T, num_fg_samples=1024, num_bg_samples=512,
                             generator=torch.Generator().manual_seed(1))
    n = 1024 + 512
    assert tuple(out["inputs"].shape) == (1, n, 3) and tuple(out["inputs.body_bounds_intersections"].shape) == (1, n, 2)
    nf = out["inputs.body_bounds_intersections"][0]
    assert bool((nf[:, 0] < nf[:, 1]).all())
    assert bool(out["inputs.mask_erode"][0, :1024].all()) and not bool(out["inputs.mask_erode"][0, 1024:].any())
    assert float(out["inputs"][0, 1024:].abs().max()) == 0.0
    # the pixel behind every ray: uv = K^-1 (x, y, 1)
    px = (out["inputs.uv"][0] @ K.t())
    xs, ys = px[:, 0].round().long(), px[:, 1].round().long()
    assert bool(body[ys[:1024], xs[:1024]].all())
    torch.testing.assert_close(out["inputs"][0, :1024], img[ys[:1024], xs[:1024]])
    box = data.bound_2d_mask(bounds, K, torch.cat([R, T.reshape(3, 1)], dim=1), H, W)
    assert bool((box[ys[1024:], xs[1024:]] & (me[ys[1024:], xs[1024:]] == 0)).all())
    assert len(set(zip(xs[:1024].tolist(), ys[:1024].tolist()))) == 1024           # without replacement
    d = out["inputs.ray_dirs"][0]
    torch.testing.assert_close(d.norm(dim=-1), torch.ones(n))
    torch.testing.assert_close(out["inputs.ray_dirs_cam"][0], d)                      # R = identity
    # too few foreground pixels: the reference's np.random.choice raises, so does this
    with pytest.raises(ValueError):
        data.training_rays(img, mask, me, bounds, K, R, T, num_fg_samples=20000, num_bg_samples=16)


@pytest.mark.gpu
def test_training_item_through_a_training_step(body, monkeypatch):
    """data.training_item -> LightningModel.training_step on cuda:0: the dataset side of a training step (frame
    composition, pixel / ray sample, regularisation point sets) feeds the model's loss and every group of parameters gets a
    gradient.  The synthetic body has no triangle mesh, so the three point sets come from a stand-in with the sampler's
    shapes (the sampler itself: tests/test_mesh_query.py)."""
    from arah_release_amd import config, data
    g = golden("f9_callers.npz")
    dev = torch.device("cuda:0")
    md = {k[3:]: g[k] for k in g.files if k.startswith("md.")}
    cam = {k[4:]: g[k] for k in g.files if k.startswith("cam.")}
    H = W = 256

    def fake_samples(v, f, w, cmin, cmax, cen, reg, inside, *a, **k):
        gen = torch.Generator(device=v.device).manual_seed(0)
        out = {"points_uniform": torch.rand(1024, 3, device=v.device, generator=gen) * 2 - 1,
               "points_skinning": v[:1024].clone(), "sampled_weights": w[:1024].clone()}
        if inside:
            out["points_inside"] = (torch.rand(1024, 3, device=v.device, generator=gen) - 0.5) * 0.2
        return out

    monkeypatch.setattr(data, "training_samples", fake_samples)
    item0 = data.frame_item(md, cam, body, H, 64, device=dev, frame_idx=5, data_idx=1)    # the fixture's K is for 64 x 64
    mask = item0["inputs.image_mask"][0]
    # "segmentation": the box's pixels whose ray passes within 15 cm of a posed vertex are the body
    d, o = item0["inputs.ray_dirs"][0], item0["image.cam_loc"][0]
    v = item0["image.smpl_vertices"][0][::8]
    t = ((v[None] - o) * d[:, None]).sum(-1)
    dist = ((o + t[..., None] * d[:, None]) - v[None]).norm(dim=-1).min(dim=1)[0]
    me = torch.zeros(H, W, dtype=torch.int64, device=dev)
    me[mask] = (dist < 0.15).long()
    image = torch.rand(H, W, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    n_fg = int((me == 1).sum()) - 1024
    assert n_fg > 256, n_fg
    item = data.training_item(md, cam, body, torch.zeros(4, 3, dtype=torch.int32), image, (me > 0).long(), me, H, 64, device=dev,
                              num_fg_samples=256, num_bg_samples=256, sample_reg_surface=True, sample_inside=True,
                              frame_idx=5, data_idx=1, generator=torch.Generator(device=dev).manual_seed(4))
    assert tuple(item["inputs.ray_dirs"].shape) == (1, 512, 3) and tuple(item["image.points_uniform"].shape) == (1, 1024, 3)
    assert "inputs.image_mask" not in item and "inputs.novel_seq" not in item
    cfg = config.builtin_config("zju313")
    lm = config.get_model(cfg, mode="test", n_data_points=4)
    lm.model.load_state_dict(config.synthetic_state_dict(cfg), strict=False)
    lm = lm.to(dev).train()
    lm.model.frames = []
    loss = lm.training_step(item)
    assert torch.isfinite(loss)
    loss.backward()
    groups = {"sdf_decoder": 0, "skinning_model": 0, "color_decoder": 0, "deviation_decoder": 0}
    for name, p in lm.model.named_parameters():
        for k in groups:
            if name.startswith(k) and p.grad is not None and float(p.grad.abs().sum()) > 0:
                groups[k] += 1
    assert all(vv > 0 for vv in groups.values()), groups


def test_optimizer_groups_against_reference(body):
    """LightningModel.configure_optimizers: the reference's parameter groups in the reference's order (fixture F11: learning
    rate, weight decay, number of tensors and elements per group; Adam defaults) -- an optimiser state dict of the reference
    addresses groups by position."""
    import json
    import os
    from conftest import GOLDEN
    from arah_release_amd import config, renderer
    ref = json.load(open(os.path.join(GOLDEN, "f11_optimizer_groups.json")))
    cfg = config.builtin_config("zju313")
    lm = config.get_model(cfg, mode="test", n_data_points=4)
    m = lm.model

    def groups(opt):
        return [{"lr": g["lr"], "weight_decay": g["weight_decay"], "tensors": len(g["params"]),
                 "elements": int(sum(p.numel() for p in g["params"]))} for g in opt.param_groups]

    opt = lm.configure_optimizers()
    assert groups(opt) == ref["plain"]
    for k, v in ref["plain_adam"].items():
        got = opt.defaults[k]
        assert (list(got) if isinstance(got, tuple) else got) == v, k
    # optimised SMPL parameters and cameras: two frames, two cameras (as the fixture registered them)
    kw = dict(frames=[4, 5], betas=np.zeros((1, 10), np.float32), body_model=body, cam_rots=np.zeros((2, 4), np.float32),
              cam_trans=np.zeros((2, 3), np.float32), n_data_points=4)
    for key, n in (("root_orient", 3), ("pose_body", 63), ("pose_hand", 6), ("trans", 3)):
        kw[key] = [np.zeros(n, np.float32), np.zeros(n, np.float32)]
    lm.model = renderer.MetaAvatarRender(m.sdf_decoder, m.skinning_model, m.color_decoder, m.deviation_decoder,
                                         train_cameras=True, train_smpl=True, train_latent_code=True,
                                         train_geo_latent_code=True, **kw)
    cfg["model"].update(train_smpl=True, train_cameras=True)
    assert groups(lm.configure_optimizers()) == ref["smpl_cameras"]


def test_test_epoch_end_writes_the_references_files(tmp_path):
    """lightning_model.py:351-401: rgb_ / normal_ / front_ / back_%06d.png under <out_dir>/vis, uint8 by truncation of
    value * 255, an existing vis directory replaced; a frame-sharded rank writes its own indices."""
    from PIL import Image
    from arah_release_amd import config
    cfg = config.builtin_config("zju313")
    cfg["training"]["out_dir"] = str(tmp_path / "run")
    lm = config.get_model(cfg, mode="test", n_data_points=2)
    g = torch.Generator().manual_seed(0)
    outs = [{k: torch.rand(3, 8, 6, generator=g) for k in ("rgb_pred", "normal_pred", "normal_front", "normal_back")} for _ in range(3)]
    os.makedirs(tmp_path / "run" / "vis")
    (tmp_path / "run" / "vis" / "stale.png").write_bytes(b"x")
    files = lm.test_epoch_end(outs)
    assert len(files) == 12 and not (tmp_path / "run" / "vis" / "stale.png").exists()
    assert sorted(os.listdir(tmp_path / "run" / "vis"))[:3] == ["back_000000.png", "back_000001.png", "back_000002.png"]
    img = np.asarray(Image.open(tmp_path / "run" / "vis" / "rgb_000001.png"))
    want = (outs[1]["rgb_pred"].permute(1, 2, 0).numpy() * 255.0).astype(np.uint8)
    assert img.shape == (8, 6, 3)
    np.testing.assert_array_equal(img, want)
    # rank 1 of 2: frames 1, 3, 5 -- and it must not wipe rank 0's files
    lm.test_epoch_end(outs, first_index=1, index_stride=2, clear=False)
    names = set(os.listdir(tmp_path / "run" / "vis"))
    assert {"rgb_000003.png", "rgb_000005.png", "rgb_000000.png", "rgb_000002.png"} <= names


@pytest.mark.gpu
def test_test_step_on_a_dataset_item(body):
    """LightningModel.test_step(item) as the reference's trainer.test calls it (lightning_model.py:306-338): the image and
    the three normal maps of the canonical mesh, channels first."""
    from arah_release_amd import config, data
    g = golden("f9_callers.npz")
    dev = torch.device("cuda:0")
    md = {k[3:]: g[k] for k in g.files if k.startswith("md.")}
    cam = {k[4:]: g[k] for k in g.files if k.startswith("cam.")}
    cfg = config.builtin_config("zju313")
    lm = config.get_model(cfg, mode="test", n_data_points=4)
    lm.model.load_state_dict(config.synthetic_state_dict(cfg), strict=False)
    lm = lm.to(dev).eval()
    lm.model.frames = []
    item = data.frame_item(md, cam, body, 512, 64, device=dev, frame_idx=5, data_idx=1)   # the fixture's K is for 64 x 64
    out = lm.test_step(item)
    assert set(out) == {"rgb_pred", "normal_pred", "normal_front", "normal_back"}
    for k, v in out.items():
        assert tuple(v.shape) == (3, 512, 512) and bool(torch.isfinite(v).all()), k
        assert 0.0 <= float(v.min()) and float(v.max()) <= 1.0 + 1e-6
    mask = item["inputs.image_mask"][0]
    assert float(out["rgb_pred"][:, ~mask].abs().max()) == 0.0           # outside the projected box nothing is written
    assert float(out["rgb_pred"].sum()) > 0.0


def _write_sequence(root, scene, n_frames=4):
    """<root>/data/odp/CoreView_000/{cam_params.json, seq/*.npz} in the reference's formats."""
    sub = os.path.join(root, "data", "odp", "CoreView_000")
    os.makedirs(os.path.join(sub, "seq"))
    rng = np.random.RandomState(0)
    for f in range(n_frames):
        fr = scene.frame(f)
        np.savez(os.path.join(sub, "seq", "%06d.npz" % f), minimal_shape=scene.verts_cano, betas=np.zeros((1, 10), np.float32),
                 Jtr_posed=fr["joints_posed"], bone_transforms=fr["bone_transforms"], trans=np.array([0.0, 0.0, 3.0], np.float32),
                 root_orient=rng.randn(3).astype(np.float32) * 0.1, pose_body=rng.randn(63).astype(np.float32) * 0.1,
                 pose_hand=np.zeros(6, np.float32))
    K = [[1228.8, 0, 512], [0, 1228.8, 512], [0, 0, 1]]
    cams = {"all_cam_names": ["1", "2"], "1": {"K": K, "D": [0] * 5, "R": np.eye(3).tolist(), "T": [[0], [0], [0.2]]},
            "2": {"K": K, "D": [0] * 5, "R": np.eye(3).tolist(), "T": [[0.1], [0], [0.3]]}}
    with open(os.path.join(sub, "cam_params.json"), "w") as f:
        json.dump(cams, f)
    return sub


def test_sequence_dataset_and_cli_surface(tmp_path, scene, body):
    """The test dataset of test.py (ZJUMOCAPODPDataset: camera-major enumeration, frame slicing, the fields get_model reads)
    and the command line of arah_release_amd.test_sequence: test.py's arguments and its overrides of the configuration."""
    from arah_release_amd import data, test_sequence
    _write_sequence(str(tmp_path), scene)
    cfg = {"data": {"dataset": "zju_mocap", "path": "elsewhere", "pose_dir": "x", "box_margin": 0.05, "test_split": ["CoreView_000"],
                    "test_views": ["9"], "test_subsampling_rate": 1, "test_start_frame": 0, "test_end_frame": 0}}
    args = test_sequence.build_parser().parse_args(["cfg.yaml", "--pose-dir", "seq", "--test-views", "2,1", "--subsampling-rate", "2",
                                                   "--start-frame", "1"])
    cfg = test_sequence.apply_overrides(cfg, args)
    assert cfg["data"]["dataset"] == "zju_mocap_odp" and cfg["data"]["path"] == "data/odp" and cfg["data"]["test_views"] == ["2", "1"]
    cfg["data"]["path"] = os.path.join(str(tmp_path), "data", "odp")
    ds = data.get_dataset("test", cfg, body)
    assert ds.cam_names == ["2", "1"] and len(ds) == 4
    assert [(d["cam_idx"], d["frame_idx"], d["data_idx"]) for d in ds.data] == [(0, 1, 0), (0, 3, 1), (1, 1, 0), (1, 3, 1)]
    item = ds.item(2, "cpu")
    assert tuple(item["inputs.image_mask"].shape) == (1, 512, 512) and int(item["inputs.cam_idx"]) == 1
    np.testing.assert_allclose(item["image.K"][0, 0, 0].item(), 1228.8 / 1024 * 512, rtol=1e-6)     # 1024 -> 512 images
    with pytest.raises(KeyError):
        cfg["data"]["test_views"] = ["5"]
        data.get_dataset("test", cfg, body)
    cfg["data"]["test_views"] = []
    assert data.get_dataset("test", cfg, body).cam_names == ["1", "2"]
    cfg["data"]["pose_dir"] = "missing"
    with pytest.raises(FileNotFoundError):
        data.get_dataset("test", cfg, body)
    cfg["data"]["dataset"] = "zju_mocap"
    with pytest.raises(ValueError):
        data.get_dataset("test", cfg, body)


@pytest.mark.gpu
def test_test_sequence_end_to_end(tmp_path, scene, body):
    """python -m arah_release_amd.test_sequence on a synthetic subject written in the reference's on-disk formats: the
    checkpoint under <out_dir>/checkpoints/last.ckpt, two cameras x two frames, four PNGs per frame under <out_dir>/vis."""
    import yaml
    from PIL import Image
    from arah_release_amd import config, test_sequence
    _write_sequence(str(tmp_path), scene, n_frames=3)
    cfg = config.builtin_config("zju313")
    cfg["training"]["out_dir"] = str(tmp_path / "out")
    cfg["data"] = {"dataset": "zju_mocap", "path": "unused", "pose_dir": "unused", "box_margin": 0.05,
                   "test_split": ["CoreView_000"], "test_views": [], "test_subsampling_rate": 1, "test_start_frame": 0,
                   "test_end_frame": 0}
    os.makedirs(tmp_path / "out" / "checkpoints")
    sd = {"model." + k: v for k, v in config.synthetic_state_dict(cfg).items()}
    sd["model.latent.weight"] = torch.zeros(4, cfg["model"]["latent_dim"])
    torch.save({"state_dict": sd}, tmp_path / "out" / "checkpoints" / "last.ckpt")
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    with pytest.raises(FileNotFoundError):
        bad = dict(cfg, training=dict(cfg["training"], out_dir=str(tmp_path / "nowhere")))
        (tmp_path / "bad.yaml").write_text(yaml.safe_dump(bad))
        test_sequence.main([str(tmp_path / "bad.yaml"), "--default-config", str(tmp_path / "bad.yaml")], body=body)
    cwd = os.getcwd()
    os.chdir(tmp_path)             # test.py overrides cfg['data']['path'] with the relative 'data/odp'
    try:
        test_sequence.main([str(tmp_path / "cfg.yaml"), "--default-config", str(tmp_path / "cfg.yaml"), "--pose-dir", "seq",
                            "--test-views", "1,2", "--subsampling-rate", "2"], body=body)
    finally:
        os.chdir(cwd)
    vis = tmp_path / "out" / "vis"
    names = sorted(os.listdir(vis))
    assert len(names) == 16 and names[0] == "back_000000.png" and "rgb_000003.png" in names      # 2 cameras x frames 0, 2
    rgb = np.asarray(Image.open(vis / "rgb_000001.png"))
    front = np.asarray(Image.open(vis / "front_000001.png"))
    assert rgb.shape == (512, 512, 3) and rgb.max() > 0
    assert front.shape == (512, 512, 3) and len(np.unique(front.reshape(-1, 3), axis=0)) > 100


def test_validation_step_against_reference():
    """LightningModel.validation_step around a stub model against the reference's own (fixture F14): image scatter, normal
    map from the surface points incl. the NaN handling, ground-truth image, PSNR; SSIM / LPIPS through caller-supplied
    callables."""
    from arah_release_amd import config
    g = golden("f14_validation_step.npz")
    cfg = config.builtin_config("zju313")
    lm = config.get_model(cfg, mode="test", n_data_points=2)
    outputs = {"rgb_values": T(g["rgb_values"]), "points_cam": T(g["points_cam"])}

    class Stub(torch.nn.Module):
        def forward(self, inputs, gen_cano_mesh=False, eval=True):
            return dict(outputs)

    lm.model = Stub()
    mask = torch.from_numpy(g["image_mask"])
    lm.compose_inputs = lambda data, eval: {"image_mask": mask}
    batch = {"inputs.image_mask": mask, "inputs": T(g["inputs"]), "inputs.img_height": torch.tensor([int(g["H"])]),
             "inputs.img_width": torch.tensor([int(g["W"])])}
    res = lm.validation_step(batch, 0, ssim_fn=lambda a, b, m: 0.5, lpips_fn=lambda a, b, m: 0.25)
    assert set(res) == {"psnr", "ssim", "lpips", "rgb_pred", "normal_pred", "rgb_gt"}
    np.testing.assert_allclose(res["psnr"], float(g["psnr"]), rtol=1e-6)
    assert res["ssim"] == float(g["ssim"]) and res["lpips"] == float(g["lpips"])
    for k in ("rgb_pred", "rgb_gt"):
        np.testing.assert_array_equal(res[k].numpy(), g[k])
    np.testing.assert_allclose(res["normal_pred"].numpy(), g["normal_pred"], rtol=0, atol=1e-6)
    assert "ssim" not in lm.validation_step(batch, 0)


@pytest.mark.gpu
def test_validation_step_on_a_dataset_item(body):
    """validation_step on the device with the real model: keys and shapes of the reference's eval dict, PSNR against a
    ground truth that IS the prediction (infinite) and against black."""
    from arah_release_amd import config, data
    g = golden("f9_callers.npz")
    dev = torch.device("cuda:0")
    md = {k[3:]: g[k] for k in g.files if k.startswith("md.")}
    cam = {k[4:]: g[k] for k in g.files if k.startswith("cam.")}
    cfg = config.builtin_config("zju313")
    lm = config.get_model(cfg, mode="test", n_data_points=4)
    lm.model.load_state_dict(config.synthetic_state_dict(cfg), strict=False)
    lm = lm.to(dev).eval()
    lm.model.frames = []
    item = data.frame_item(md, cam, body, 128, 64, device=dev, frame_idx=5, data_idx=1)
    res = lm.validation_step(item)                                       # frame_item's 'inputs' are black pixels
    assert set(res) == {"psnr", "rgb_pred", "normal_pred", "rgb_gt"}
    for k in ("rgb_pred", "normal_pred", "rgb_gt"):
        assert tuple(res[k].shape) == (3, 128, 128) and bool(torch.isfinite(res[k]).all()), k
    assert float(res["rgb_gt"].abs().max()) == 0.0 and np.isfinite(res["psnr"]) and res["psnr"] > 0
    mask = item["inputs.image_mask"][0]
    item["inputs"] = res["rgb_pred"].permute(1, 2, 0)[mask].unsqueeze(0).clone()
    again = lm.validation_step(item)
    assert again["psnr"] == float("inf") and torch.equal(again["rgb_gt"], again["rgb_pred"])
    # the normal map: unit-length where defined, (0, 0, 0) where the finite differences were NaN
    n = again["normal_pred"].permute(1, 2, 0) * 2 - 1
    defined = (again["normal_pred"] != 0).any(0)
    assert float((n[defined].norm(dim=-1) - 1).abs().max()) < 1e-3
