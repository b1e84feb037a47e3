"""Pin the oracle (oracle/arah_oracle.py) against vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only.

Tolerances: element-wise fixtures rtol 1e-4 / atol 1e-5; path-dependent fixtures
(tracer, whole forward): mask agreement >= 99.5 %, values compared on agreeing entries,
PSNR(oracle, reference) >= 45 dB with the reference's own PSNR formula.
"""
import numpy as np
import pytest
import torch

from conftest import golden, psnr, get_model
from oracle import arah_oracle as O

RTOL, ATOL = 1e-4, 1e-5


def T(x):
    return torch.from_numpy(np.asarray(x))


@pytest.fixture(scope="module")
def frame377(scene):
    model, cfg = get_model("zju377_mono")
    inputs = scene.make_inputs(64, 64, frame_idx=0)
    return O.frame_from_model(model, inputs)


def test_f2_pointwise(frame377):
    g = golden("f2_pointwise.npz")
    fr = frame377
    assert abs(fr.coord_min - float(g["coord_min"][0])) == 0 and abs(fr.coord_max - float(g["coord_max"][0])) == 0
    np.testing.assert_allclose(O.hierarchical_softmax(T(g["logits"])).numpy(), g["hsoftmax"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(O.unnormalize_points(fr, T(g["x_norm"])).numpy(), g["x_hat"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(O.normalize_points(fr, T(g["x_hat"])).numpy(), g["x_norm_back"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(O.skin_logits(fr, T(g["x_norm"])).numpy(), g["deformer_logits"], rtol=RTOL, atol=1e-4)
    np.testing.assert_allclose(O.query_weights(fr, T(g["x_hat"])).numpy(), g["weights"], rtol=1e-3, atol=ATOL)
    xb, Tm = O.lbs_forward(fr, T(g["x_hat"]))
    np.testing.assert_allclose(xb.numpy(), g["x_bar"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(Tm.numpy(), g["T"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(O.lbs_jacobian(fr, T(g["x_hat"])).numpy(), g["jac"], rtol=1e-3, atol=1e-4)


def test_f3_sdf(frame377):
    g = golden("f3_sdf.npz")
    sdf, feat = O.sdf_forward(frame377, T(g["x_norm"]))
    np.testing.assert_allclose(sdf.numpy(), g["sdf"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(feat.numpy(), g["feat"], rtol=RTOL, atol=ATOL)
    sdf2, feat2, grad = O.sdf_forward_grad(frame377, T(g["x_norm"]))
    np.testing.assert_allclose(sdf2.numpy(), g["sdf"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(grad.numpy(), g["grad"], rtol=1e-3, atol=1e-4)


def test_f1_broyden3(frame377):
    g = golden("f1_broyden3.npz")
    fr = frame377
    tgt = T(g["tgt"])

    def resid(x, k):
        xb, Tm = O.lbs_forward(fr, x)
        return xb - tgt[k], Tm

    x, Tm, err, ok = O.broyden(resid, T(g["x0"]), T(g["T0"]), T(g["Jinv0"]))
    valid = g["valid"]
    assert (ok.numpy() == valid).mean() >= 0.995
    both = ok.numpy() & valid
    assert both.sum() > 200
    np.testing.assert_allclose(x.numpy()[both], g["result"][both], rtol=0, atol=2e-5)
    np.testing.assert_allclose(Tm.numpy()[both], g["transforms"][both], rtol=1e-3, atol=1e-4)
    # points whose start was never improved keep the caller's T_init (broyden.py:41, 57-61)
    never = np.isclose(g["transforms"][:, 0, 0], 7.0)
    assert (np.isclose(Tm.numpy()[:, 0, 0], 7.0) == never).all()
    np.testing.assert_allclose(err.numpy()[~valid], g["diff"][~valid], rtol=5e-2, atol=1e-4)


def test_f1_broyden4(frame377):
    """D = 4: the joint root find on (x_hat, depth) -- Jacobian assembly, residual and broyden() -- against the
    reference's search_iso_surface_depth (RFU:365-484) on 256 rays with perturbed starts, masked-out rays and a
    few hopeless starts."""
    g = golden("f1_broyden4.npz")
    fr = frame377
    valid = torch.from_numpy(g["valid"])
    x, z, Tm, conv = O.joint_root_find(fr, T(g["cam"]), T(g["rays"]), valid, T(g["x0"]), T(g["z0"]), T(g["T0"]))
    ref_conv = g["converged"]
    assert (conv.numpy() == ref_conv).mean() >= 0.995
    both = conv.numpy() & ref_conv
    assert both.sum() > 200
    np.testing.assert_allclose(x.numpy()[both], g["x_opt"][both], rtol=0, atol=2e-5)
    np.testing.assert_allclose(z.numpy()[both], g["z_opt"][both], rtol=0, atol=2e-5)
    np.testing.assert_allclose(Tm.numpy()[both], g["T_opt"][both], rtol=1e-3, atol=1e-4)
    off = ~g["valid"]                               # rays outside the mask keep their inputs (RFU:472-482)
    np.testing.assert_array_equal(x.numpy()[off], g["x0"][off])
    np.testing.assert_array_equal(z.numpy()[off], g["z0"][off])
    np.testing.assert_array_equal(Tm.numpy()[off], g["T0"][off])
    assert not conv.numpy()[off].any() and not ref_conv[off].any()


@pytest.mark.parametrize("name", ["zju377_mono", "zju313"])
def test_f4_color(scene, name):
    g = golden("f4_color_%s.npz" % name)
    model, cfg = get_model(name)
    fr = O.frame_from_model(model, scene.make_inputs(64, 64, frame_idx=0))
    rgb = O.color_forward(fr, T(g["points"]), T(g["normals"]), T(g["view"]), T(g["feat"]))
    np.testing.assert_allclose(rgb.numpy(), g["rgb"], rtol=RTOL, atol=ATOL)


def _tracer_inputs(scene, g):
    inputs = scene.make_inputs(int(g["H"]), int(g["W"]), frame_idx=int(g["frame_idx"]), max_rays=int(g["max_rays"]))
    o = inputs["cam_loc"].expand(inputs["ray_dirs"].shape[1], 3).contiguous()
    d = inputs["ray_dirs"][0]
    nf = inputs["body_bounds_intersections"][0]
    return inputs, o, d, nf[:, 0].contiguous(), nf[:, 1].contiguous()


@pytest.mark.parametrize("tag", ["s64", "s32"])
def test_f5_tracer(scene, tag):
    g = golden("f5_tracer_%s.npz" % tag)
    model, cfg = get_model("zju377_mono")
    inputs, o, d, near, far = _tracer_inputs(scene, g)
    fr = O.frame_from_model(model, inputs)
    S, nn, nfar = int(g["n_steps"]), int(g["n_near"]), int(g["n_far"])
    xn, Ts, conv, start, end = O.trace_rays(fr, o, d, near, far)
    ref_conv = g["network_body_mask"]
    assert (conv.numpy() == ref_conv).mean() >= 0.995
    both = conv.numpy() & ref_conv
    assert both.sum() > 50
    np.testing.assert_allclose(start.numpy()[both], g["dists"][both], rtol=0, atol=1e-4)
    np.testing.assert_allclose(xn.numpy()[both], g["points_hat_norm"][both], rtol=0, atol=2e-4)
    np.testing.assert_allclose(start.numpy()[~both & ~ref_conv & ~conv.numpy()],
                               g["dists"][~both & ~ref_conv & ~conv.numpy()], rtol=0, atol=0)
    # sampler on the reference's own ray classification, so that depth samples are comparable 1:1
    spts, sT, smask, sz = O.sample_and_canonicalize(fr, o, d, T(ref_conv), T(g["dists"]), far, near, S, nn, nfar)
    np.testing.assert_array_equal(sz.numpy(), g["sampler_dists"])      # bit-exact depth samples
    ref_mask = g["sampler_converge_mask"]
    assert (smask.numpy() == ref_mask).mean() >= 0.995
    bm = smask.numpy() & ref_mask
    np.testing.assert_allclose(spts.numpy()[bm], g["sampler_pts"][bm], rtol=0, atol=2e-4)
    np.testing.assert_allclose(sT.numpy()[bm][:, :3, :].reshape(-1, 12), g["sampler_transforms34"][bm], rtol=1e-3,
                               atol=2e-4)
    # masked-off sample slots (beyond near+far+1 on converged rays) are zeros on both sides (RT:549-553)
    off = np.abs(g["sampler_pts"]).sum(-1) == 0
    assert off.sum() > 0 and np.abs(spts.numpy()[off]).max() == 0 and not smask.numpy()[off].any()


@pytest.mark.parametrize("name,tag", [("zju377_mono", "s64"), ("h36m", "s64"), ("zju377_mono", "s32")])
def test_f6_shade(scene, name, tag):
    g5 = golden("f5_tracer_%s.npz" % tag)
    g = golden("f6_shade_%s_%s.npz" % (name, tag))
    model, cfg = get_model(name)
    inputs, o, d, near, far = _tracer_inputs(scene, g5)
    fr = O.frame_from_model(model, inputs)
    vol = g["vol_mask"]
    T34 = g5["sampler_transforms34"].reshape(-1, int(g["n_steps"]), 3, 4)
    T44 = np.concatenate([T34, np.tile(np.array([0, 0, 0, 1], np.float32), T34.shape[:2] + (1, 1))], axis=2)
    rgb, acc = O.shade_composite(fr, T(g5["sampler_pts"][vol]), T(g5["sampler_dists"][vol]), T(T44[vol]),
                                 T(g5["sampler_converge_mask"][vol]), d[T(vol)], int(g["n_steps"]),
                                 cfg["model"]["cano_view_dirs"])
    np.testing.assert_allclose(rgb.numpy(), g["rgb"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(acc.numpy(), g["acc"], rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("fname,name", [("f7_forward_zju377_mono_64x64_s64.npz", "zju377_mono"),
                                        ("f7_forward_zju313_64x64_s64.npz", "zju313"),
                                        ("f7_forward_h36m_48x48_s32.npz", "h36m"),
                                        ("f7_forward_zju377_mono_128x128_s32.npz", "zju377_mono"),
                                        ("f7_forward_h36m_40x40_s128.npz", "h36m")])   # BASELINE config 5's sampling (128, 32, 32)
def test_f7_forward(scene, fname, name):
    g = golden(fname)
    model, cfg = get_model(name)
    inputs = scene.make_inputs(int(g["H"]), int(g["W"]), frame_idx=int(g["frame_idx"]))
    np.testing.assert_array_equal(inputs["ray_dirs"][0].numpy(), g["ray_dirs"])   # same rays as the reference saw
    out = O.render_inputs(model, inputs, cfg["model"]["cano_view_dirs"], int(g["n_steps"]), int(g["n_near"]),
                          int(g["n_far"]))
    assert (out["network_body_mask"].numpy() == g["network_body_mask"]).mean() >= 0.995
    assert psnr(out["rgb_values"].numpy(), g["rgb_values"]) >= 45.0
    hit_ref = np.abs(g["points_cam"]).sum(-1) > 0
    hit = np.abs(out["points_cam"].numpy()).sum(-1) > 0
    assert (hit == hit_ref).mean() >= 0.995
    both = hit & hit_ref
    np.testing.assert_allclose(out["points_cam"].numpy()[both], g["points_cam"][both], rtol=0, atol=2e-4)


@pytest.mark.parametrize("fname,n_sub,name", [("f7_forward_zju377_mono_256x256_s32.npz", 2048, "zju377_mono"),    # BASELINE config 1 at its full size
                                              ("f7_forward_zju377_mono_512x512_s64.npz", 1024, "zju377_mono"),    # BASELINE config 2: the benchmark frame
                                              ("f7_forward_h36m_128x128_s128.npz", 512, "h36m")])   # round 6: config 5's shapes and sampling (128, 32, 32)
def test_f7_full_frames_on_a_ray_subset(scene, fname, n_sub, name):
    """The reference's own render of BASELINE configs 1 and 2 at FULL size (tests/golden/make_golden.py f7full: 256 x 256 x 32 and
    the 512 x 512 x 64 benchmark frame, one and nine minutes of the reference on eight cores).  The oracle renders an evenly
    spread subset of the frame's rays -- rays are independent, the subset is what make_inputs(max_rays=...) picks -- and must
    agree with the reference's rows for those rays; the GPU test holds the HIP path to the whole frame."""
    g = golden(fname)
    model, cfg = get_model(name)
    H, W, fidx = int(g["H"]), int(g["W"]), int(g["frame_idx"])
    n_full = int(g["rgb_values"].shape[0])
    inputs = scene.make_inputs(H, W, frame_idx=fidx, max_rays=n_sub)
    sel = np.linspace(0, n_full - 1, n_sub).round().astype(int)     # synthetic.make_inputs' own subsampling rule
    out = O.render_inputs(model, inputs, cfg["model"]["cano_view_dirs"], int(g["n_steps"]), int(g["n_near"]), int(g["n_far"]))
    assert (out["network_body_mask"].numpy() == g["network_body_mask"][sel]).mean() >= 0.995
    assert psnr(out["rgb_values"].numpy(), g["rgb_values"][sel]) >= 45.0
    hit_ref = np.abs(g["points_cam"][sel]).sum(-1) > 0
    hit = np.abs(out["points_cam"].numpy()).sum(-1) > 0
    assert (hit == hit_ref).mean() >= 0.995
    both = hit & hit_ref
    assert (np.abs(out["points_cam"].numpy()[both] - g["points_cam"][sel][both]).max(-1) <= 2e-4).mean() >= 0.999


@pytest.mark.parametrize("tag", ["s64", "s32"])
def test_f19_training_time_depth_jitter(tag):
    """The oracle's stratified jitter (perturb_z_vals / the training branches of ray_sampler, RT:298-350) against the
    reference's own z_vals for its own recorded torch.rand draws (f19: rays with and without a surface hit, surfaces closer
    to the near bound than the surface range)."""
    g = golden("f19_jitter_depths.npz")
    S, n_near, n_far = [int(v) for v in g[tag + ".cfg"]]
    t = lambda k: torch.from_numpy(np.asarray(g[tag + "." + k]))
    z, mask = O.sample_depths(t("conv").bool(), t("start"), t("end"), t("near"), S, n_near, n_far,
                              jitter=(t("rand_steps"), t("rand_near"), t("rand_far")))
    ref = g[tag + ".z"]
    assert z.shape == ref.shape
    np.testing.assert_allclose(z.numpy(), ref, rtol=0, atol=1e-6)
    conv = g[tag + ".conv"].astype(bool)
    assert bool(mask[~conv].all()) and int(mask[conv].sum(-1).unique()) == n_near + 1 + n_far
    # eval mode is the same call without draws and differs from it
    z0, _ = O.sample_depths(t("conv").bool(), t("start"), t("end"), t("near"), S, n_near, n_far)
    assert float((z0 - z).abs().max()) > 1e-3

