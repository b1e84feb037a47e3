"""Image preparation of a training item (arah_release_amd/imageops.py; the reference uses OpenCV, zju_mocap.py:209-262) and
the training dataset built on it.  OpenCV is absent: the operations are checked against their definitions."""
import json
import os

import numpy as np
import pytest
import torch

from arah_release_amd import imageops


def test_erode_dilate_and_rim():
    rng = np.random.RandomState(0)
    m = (rng.rand(17, 23) > 0.6).astype(np.int64)
    m[5:12, 6:16] = 1
    er, di = imageops.erode(torch.from_numpy(m)).numpy(), imageops.dilate(torch.from_numpy(m)).numpy()
    H, W = m.shape
    for i in range(H):
        for j in range(W):
            win = m[max(0, i - 2):i + 3, max(0, j - 2):j + 3]              # outside pixels do not take part
            assert er[i, j] == win.min() and di[i, j] == win.max(), (i, j)
    rim = imageops.rim_mask(torch.from_numpy(m * 255)).numpy()
    assert set(np.unique(rim)) <= {0, 1, 100}
    np.testing.assert_array_equal(rim == 100, (di - er) == 1)
    np.testing.assert_array_equal(rim == 1, er == 1)
    np.testing.assert_array_equal(imageops.rim_mask(torch.from_numpy(m * 255), erode_mask=False).numpy(), m)


def test_undistort_maps_and_interpolates():
    H, W = 40, 52
    K = torch.tensor([[60.0, 0, 25.5], [0, 58.0, 19.0], [0, 0, 1]])
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    ramp = (0.7 * u + 1.3 * v + 2.0).float()
    img = torch.stack([ramp, 2 * ramp, ramp * 0 + 5], -1)
    assert torch.equal(imageops.undistort(img, K, [0, 0, 0, 0, 0]), img)
    D = [-0.12, 0.03, 0.002, -0.001, 0.005]
    out = imageops.undistort(img, K, D)
    xn, yn = (u - 25.5) / 60.0, (v - 19.0) / 58.0
    r2 = xn ** 2 + yn ** 2
    rad = 1 + D[0] * r2 + D[1] * r2 ** 2 + D[4] * r2 ** 3
    us = (xn * rad + 2 * D[2] * xn * yn + D[3] * (r2 + 2 * xn ** 2)) * 60.0 + 25.5
    vs = (yn * rad + D[2] * (r2 + 2 * yn ** 2) + 2 * D[3] * xn * yn) * 58.0 + 19.0
    inside = (us >= 0) & (us <= W - 1) & (vs >= 0) & (vs <= H - 1)
    want = (0.7 * us + 1.3 * vs + 2.0).float()                     # bilinear interpolation reproduces a linear ramp
    torch.testing.assert_close(out[..., 0][inside], want[inside], rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(out[..., 2][inside], torch.full_like(want[inside], 5.0), rtol=1e-6, atol=1e-5)
    assert inside.float().mean() > 0.9 and float((us - u).abs().max()) > 0.5     # the distortion does move pixels
    msk = (ramp > 40).to(torch.uint8) * 255
    um = imageops.undistort(msk, K, D)
    assert um.dtype == torch.uint8 and um.shape == msk.shape and set(np.unique(um.numpy())) != {0}


def test_resize_conventions():
    x = torch.arange(8.0)[None, :].repeat(6, 1)
    img = torch.stack([x, 3 * x + 1], -1)
    small = imageops.resize_linear(img, (3, 4))
    torch.testing.assert_close(small[..., 0], torch.tensor([0.5, 2.5, 4.5, 6.5])[None].repeat(3, 1))      # centres at half integers
    torch.testing.assert_close(small[..., 1], 3 * small[..., 0] + 1)
    big = imageops.resize_linear(img, (6, 16))
    torch.testing.assert_close(big[0, :3, 0], torch.tensor([0.0, 0.25, 0.75]))     # src = (x + 0.5) / 2 - 0.5, clamped at the edge
    m = torch.arange(35).reshape(5, 7)
    near = imageops.resize_nearest(m, (3, 4))
    ys, xs = [0, 1, 3], [0, 1, 3, 5]                                                                       # floor(dst * src / dst_size)
    np.testing.assert_array_equal(near.numpy(), m.numpy()[np.ix_(ys, xs)])
    assert torch.equal(imageops.resize_nearest(m, (5, 7)), m)


def test_training_dataset_item_on_the_host(tmp_path, scene, monkeypatch):
    """ZJUMOCAPDataset (train) on files written in the reference's layout: enumeration, and one item end to end on the CPU
    (the regularisation point sets need the GPU's mesh query: stand-in with the sampler's shapes)."""
    from PIL import Image
    from arah_release_amd import data, smpl
    body = smpl.BodyModel.synthetic(scene)
    sub = tmp_path / "CoreView_000"
    (sub / "models").mkdir(parents=True)
    H = W = 256
    K = [[300.0, 0, 128], [0, 300.0, 128], [0, 0, 1]]
    cams = {"all_cam_names": ["1", "2"]}
    for c in ("1", "2"):
        cams[c] = {"K": K, "D": [0.0] * 5 if c == "1" else [-0.05, 0.01, 0, 0, 0], "R": np.eye(3).tolist(), "T": [[0], [0], [0.2]]}
        (sub / c).mkdir()
    rng = np.random.RandomState(0)
    for f in range(3):
        fr = scene.frame(f)
        np.savez(sub / "models" / ("%06d.npz" % f), minimal_shape=scene.verts_cano, betas=np.zeros((1, 10), np.float32),
                 Jtr_posed=fr["joints_posed"], bone_transforms=fr["bone_transforms"], trans=np.array([0.0, 0.0, 3.0], np.float32),
                 root_orient=np.zeros(3, np.float32), pose_body=np.zeros(63, np.float32), pose_hand=np.zeros(6, np.float32))
        # "photograph": the projected posed vertices splatted into a silhouette
        v = fr["smpl_verts"] + np.array([0, 0, 0.2], np.float32)
        px = np.round(v[:, :2] / v[:, 2:3] * 300.0 + 128).astype(int)
        sil = np.zeros((H, W), np.uint8)
        ok = (px[:, 0] >= 2) & (px[:, 0] < W - 2) & (px[:, 1] >= 2) & (px[:, 1] < H - 2)
        for dx in range(-3, 4):
            for dy in range(-3, 4):
                sil[px[ok, 1] + dy, px[ok, 0] + dx] = 255
        for c in ("1", "2"):
            Image.fromarray(rng.randint(0, 255, (H, W, 3)).astype(np.uint8)).save(sub / c / ("%06d.jpg" % f))
            Image.fromarray(sil).save(sub / c / ("%06d.png" % f))
    (sub / "cam_params.json").write_text(json.dumps(cams))

    def fake_samples(v, f, w, cmin, cmax, cen, reg, inside, *a, **k):
        return {"points_uniform": torch.zeros(1024, 3), "points_skinning": v[:24].clone(), "sampled_weights": torch.eye(24)}

    monkeypatch.setattr(data, "training_samples", fake_samples)
    ds = data.TrainingDataset(str(tmp_path), subjects=["CoreView_000"], img_size=(256, 256), num_fg_samples=256, num_bg_samples=128,
                              sampling_rate=2, body=body, faces=np.zeros((1, 3), np.int32))
    assert len(ds) == 4 and [(d["cam_idx"], d["frame_idx"], d["data_idx"]) for d in ds.data] == [(0, 0, 0), (0, 2, 1), (1, 0, 0), (1, 2, 1)]
    assert ds.data[3]["img_file"].endswith(os.path.join("2", "000002.jpg")) and ds.data[3]["mask_file"].endswith("000002.png")
    for idx in (1, 3):                                            # camera 1: no distortion; camera 2: distorted
        item = ds.item(idx, "cpu", generator=torch.Generator().manual_seed(idx))
        assert tuple(item["inputs"].shape) == (1, 384, 3) and tuple(item["inputs.ray_dirs"].shape) == (1, 384, 3)
        assert bool(item["inputs.mask_erode"][0, :256].all()) and not bool(item["inputs.mask_erode"][0, 256:].any())
        assert float(item["inputs"][0, :256].max()) <= 1.0 and float(item["inputs"][0, :256].mean()) > 0.2
        assert float(item["inputs"][0, 256:].abs().max()) == 0.0
        nf = item["inputs.body_bounds_intersections"][0]
        assert bool((nf[:, 0] < nf[:, 1]).all()) and "image.points_uniform" in item and int(item["inputs.frame_idx"]) == 2
    with pytest.raises(ValueError):
        data.TrainingDataset(str(tmp_path), subjects=["CoreView_000"], sampling="patch", body=body, faces=np.zeros((1, 3), np.int32))
    (sub / "2" / "000002.png").unlink()
    with pytest.raises(AssertionError):
        data.TrainingDataset(str(tmp_path), subjects=["CoreView_000"], body=body, faces=np.zeros((1, 3), np.int32))


def _tiny_capture(root, scene, subject_dir, img_dirs, n_frames=2, size=64):
    from PIL import Image
    os.makedirs(os.path.join(subject_dir, "models"))
    for d in set(img_dirs):
        os.makedirs(d, exist_ok=True)
    for f in range(n_frames):
        fr = scene.frame(f)
        np.savez(os.path.join(subject_dir, "models", "%06d.npz" % f), minimal_shape=scene.verts_cano,
                 betas=np.zeros((1, 10), np.float32), Jtr_posed=fr["joints_posed"], bone_transforms=fr["bone_transforms"],
                 trans=np.array([0.0, 0.0, 3.0], np.float32), root_orient=np.zeros(3, np.float32),
                 pose_body=np.zeros(63, np.float32), pose_hand=np.zeros(6, np.float32))
        Image.fromarray(np.zeros((size, size, 3), np.uint8)).save(os.path.join(img_dirs[0], "%06d.jpg" % f))
        Image.fromarray(np.zeros((size, size), np.uint8)).save(os.path.join(img_dirs[1], "%06d.png" % f))


def test_dataset_variants_enumerate_their_layouts(tmp_path, scene):
    """H36M (<subject>/Posing/..., rim only while training, area reduction first) and People-Snapshot (camera.pkl, image/ and
    mask/ directories, gendered subjects): the layouts of reference data/h36m.py and data/people_snapshot.py."""
    import pickle
    from arah_release_amd import data, smpl
    body = smpl.BodyModel.synthetic(scene)
    faces = np.zeros((1, 3), np.int32)
    posing = str(tmp_path / "h36m" / "S9" / "Posing")
    _tiny_capture(str(tmp_path), scene, posing, [os.path.join(posing, "54138969"), os.path.join(posing, "54138969")])
    cam = {"K": [[50.0, 0, 32], [0, 50.0, 32], [0, 0, 1]], "D": [0.0] * 5, "R": np.eye(3).tolist(), "T": [[0], [0], [0.2]]}
    with open(os.path.join(posing, "cam_params.json"), "w") as f:
        json.dump({"all_cam_names": ["54138969"], "54138969": cam}, f)
    ds = data.H36MDataset(str(tmp_path / "h36m"), subjects=["S9"], img_size=(32, 32), body=body, faces=faces)
    assert len(ds) == 2 and ds.data[1]["model_file"].endswith(os.path.join("Posing", "models", "000001.npz"))
    m = torch.zeros(16, 16, dtype=torch.uint8)
    m[4:12, 4:12] = 255
    assert int((ds._rim(m) == 100).sum()) > 0                                   # training: rim marked (erode_mask default)
    ds.mode = "val"
    assert int((ds._rim(m) == 100).sum()) == 0                                  # h36m.py:211, the opposite of zju_mocap.py:212
    img = torch.arange(64.0 * 64 * 3).reshape(64, 64, 3)
    out, mk, rim, orig = ds._prepare(img, torch.zeros(64, 64, dtype=torch.uint8), torch.zeros(64, 64, dtype=torch.int64),
                                     torch.tensor(cam["K"]), np.zeros(5))
    assert tuple(out.shape) == (32, 32, 3) and orig == (32, 32)                 # intrinsics already refer to img_size
    torch.testing.assert_close(out[0, 0] * 255.0, img[:2, :2].mean((0, 1)))     # area reduction by 2 = mean of 2 x 2
    # People-Snapshot
    sub = str(tmp_path / "ps" / "female-3-casual")
    _tiny_capture(str(tmp_path), scene, sub, [os.path.join(sub, "image"), os.path.join(sub, "mask")], size=48)
    with open(os.path.join(sub, "camera.pkl"), "wb") as f:
        pickle.dump({"camera_f": np.array([70.0, 71.0]), "camera_c": np.array([24.0, 23.0]), "camera_k": np.zeros(5),
                     "height": 48, "width": 48}, f)
    ps = data.PeopleSnapshotDataset(str(tmp_path / "ps"), subjects=["female-3-casual"], img_size=(48, 48), body=body, faces=faces)
    assert len(ps) == 2 and ps.cam_names == ["1"] and ps.data[0]["gender"] == "female" and ps.orig_img_size == (48, 48)
    K = ps.cameras["1"]["K"]
    assert K[0, 0] == 70.0 and K[1, 1] == 71.0 and K[0, 2] == 24.0 and K[1, 2] == 23.0 and K[2, 2] == 1.0
    assert ps.data[1]["img_file"].endswith(os.path.join("image", "000001.jpg")) and ps.data[1]["mask_file"].endswith(os.path.join("mask", "000001.png"))
    assert data.PeopleSnapshotDataset._gender(ps, "male-2-sport") == "male"
