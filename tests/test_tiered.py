"""Tiered evaluation of the eval forward (csrc/tier.hpp) against the untiered one, through the model entry.

The reference runs loops C and D over every depth sample of every ray (ray_tracing.py:313-380,
implicit_differentiable_renderer.py:261-396); the tiered forward skips the samples its occupancy bitmap certifies as
sigma = +0 and the rays that cannot meet the surface.  What must hold, bit for bit (integer / mask outputs) and bit for bit
(fp32 outputs: the evaluated samples take the same kernels on the same inputs):

* rgb_values, network_body_mask, points_cam of the two paths are EQUAL;
* no violation: every ray on which the exact path finds a valid sample with density > 0 went to the exact tier
  (surface ray or promoted) -- {exact sigma > 0} is a subset of {rays sent to the exact tier}.
"""
import numpy as np
import pytest
import torch

from conftest import golden

gpu = pytest.mark.gpu


def _both_ways(model, inputs, n_steps):
    """Render `inputs` untiered and tiered; returns the two output dicts, the tiers' per-ray arrays and counters."""
    idhr = model.idhr_network
    dev = inputs["ray_dirs"].device
    n = inputs["ray_dirs"].shape[1]
    keep = (idhr.tiering, idhr.adaptive_shading)
    idhr.adaptive_shading = False
    res = {}
    try:
        for name, on in (("exact", False), ("tiered", True)):
            idhr.tiering = on
            with torch.no_grad():
                ws = idhr.ray_tracer.workspace(dev)
                if ws.buf is not None:
                    ws.reset_counters()
                out = model(dict(inputs), eval=True)
                ws = idhr.ray_tracer.workspace(dev)
                tier, pos = ws.tier_debug(n, n_steps)
                smp = ws.debug_samples(n, n_steps, which=("mask", "shaded"))
                torch.cuda.synchronize()
                res[name] = {"out": {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}, "tier": tier.clone(),
                             "pos": pos.clone(), "ctr": ws.counters(), "occ": ws.occupancy_info() if on else None,
                             "mask": smp["mask"].clone(), "sigma": smp["shaded"][:, 3].clone()}
    finally:
        idhr.tiering, idhr.adaptive_shading = keep
    return res


def _assert_same(res, label):
    e, t = res["exact"], res["tiered"]
    for key in ("rgb_values", "network_body_mask", "points_cam"):
        assert torch.equal(e["out"][key], t["out"][key]), "%s: %s differs on %d rays" % (
            label, key, int((e["out"][key] != t["out"][key]).reshape(e["out"][key].shape[1], -1).any(-1).sum()))
    violations = int(((e["pos"] == 1) & (t["tier"] == 0)).sum())
    assert violations == 0, "%s: %d rays with density > 0 in the exact path were certified zero" % (label, violations)
    # sample level: whatever the tiers evaluated is valid or not exactly as in the exact path, and carries the exact path's
    # density bit for bit -- in particular the samples whose density the tiers only CERTIFY as +0 (witnesses, phase 2)
    ev = t["mask"] == 1
    assert bool((e["mask"][ev] == 1).all()), label
    assert torch.equal(e["sigma"][ev], t["sigma"][ev]), "%s: %d evaluated samples differ in density" % (
        label, int((e["sigma"][ev] != t["sigma"][ev]).sum()))
    c = t["ctr"]
    assert c["n_tier_rays"] == e["pos"].numel()
    assert c["n_tier_rays_surface"] + c["n_tier_rays_promoted"] + c["n_tier_rays_skipped"] == c["n_tier_rays"]
    # the tiers evaluate a subset of the exact path's samples and shade exactly the same ones
    assert c["n_col"] == e["ctr"]["n_col"] and c["n_density"] <= e["ctr"]["n_density"] and c["n_canon"] <= e["ctr"]["n_canon"]
    assert c["n_density_p2"] == 0   # phase 2's samples are certified: convergence only
    return c


@gpu
@pytest.mark.parametrize("fname,name", [("f7_forward_zju377_mono_64x64_s64.npz", "zju377_mono"),
                                        ("f7_forward_zju313_64x64_s64.npz", "zju313"),
                                        ("f7_forward_h36m_48x48_s32.npz", "h36m"),
                                        ("f7_forward_zju377_mono_128x128_s32.npz", "zju377_mono"),
                                        ("f7_forward_h36m_40x40_s128.npz", "h36m"),
                                        ("f7_forward_h36m_128x128_s128.npz", "h36m"),   # round 6: config 5's shapes and sampling, 128 x 128 (the reference, 8 threads)
                                        ("f7_forward_zju377_mono_256x256_s32.npz", "zju377_mono"),
                                        ("f7_forward_zju377_mono_512x512_s64.npz", "zju377_mono")])
def test_tiers_equal_the_exact_path_on_the_reference_fixtures(scene, fname, name):
    """The inputs of every F7 fixture (the frames the reference itself rendered): both paths, equal outputs, no violation."""
    from arah_release_amd import config
    g = golden(fname)
    dev = torch.device("cuda:0")
    S = int(g["n_steps"])
    model, _ = config.build_synthetic_model(name, S, int(g["n_near"]), int(g["n_far"]), device=dev)
    inputs = scene.make_inputs(int(g["H"]), int(g["W"]), frame_idx=int(g["frame_idx"]), device=dev)
    res = _both_ways(model, inputs, S)
    c = _assert_same(res, fname)
    assert res["tiered"]["occ"]["valid"] == 1 and res["tiered"]["occ"]["overflow"] == 0
    assert 0.9 < res["tiered"]["occ"]["lip_pose"] < 32.0, res["tiered"]["occ"]   # the forward skinning's stretch, measured per cell (most 1.0-1.2, the steepest 2-10.4 where the softmax tree switches bones; a sanity range, the dilation follows the measurement)
    assert c["n_tier_samples_skipped"] > 0


@gpu
def test_tiers_equal_the_exact_path_on_the_fp32_engine(scene, monkeypatch):
    """The exact engine (ARAH_PRECISION=fp32: v_mfma_f32_16x16x4_f32 everywhere, loop C's tile kernel): the bitmap is built with
    that engine's SDF and the same certificate holds."""
    from arah_release_amd import config
    monkeypatch.setenv("ARAH_PRECISION", "fp32")
    dev = torch.device("cuda:0")
    model, _ = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
    for fi in (0, 5):
        inputs = scene.make_inputs(192, 192, frame_idx=fi, device=dev)
        c = _assert_same(_both_ways(model, inputs, 64), "fp32 engine, frame %d" % fi)
        assert c["n_tier_samples_skipped"] > c["n_tier_samples_p1"]


@gpu
def test_tiers_equal_the_exact_path_on_the_benchmark_frames(scene):
    """bench.py's workload: frames 0..19 at 512 x 512 x 64 (BASELINE config 2), every ray."""
    from arah_release_amd import config
    dev = torch.device("cuda:0")
    model, _ = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
    skipped = evaluated = 0
    for fi in range(20):
        inputs = scene.make_inputs(512, 512, frame_idx=fi, device=dev)
        c = _assert_same(_both_ways(model, inputs, 64), "frame %d" % fi)
        skipped += c["n_tier_samples_skipped"]
        evaluated += c["n_tier_samples_p1"] + c["n_tier_samples_p2"]
    assert skipped > 2 * evaluated   # the synthetic subject: ~75 % of the depth samples are never evaluated


@gpu
def test_tiers_equal_the_exact_path_on_the_wide_range_subject(scene):
    """Fixture F17's subject (skinning-MLP gains x 8: the SCALED instance of loop C's kernel, a skinning field far from the
    body's own weights): the certificate must hold for it too."""
    from arah_release_amd import config
    g = golden("f17_wide_skinning.npz")
    dev = torch.device("cuda:0")
    model, _ = config.build_synthetic_model("zju377_mono", device=dev)
    config.widen_skinning_(model, float(g["scale"]))
    for fi in (0, 3):
        inputs = scene.make_inputs(160, 160, frame_idx=fi, device=dev)
        _assert_same(_both_ways(model, inputs, 64), "wide-range subject, frame %d" % fi)


@gpu
def test_tiers_with_a_wide_band_fall_back_to_one_phase(scene):
    """beta = 3e-2: the band is 0.54 m, wider than the body.  Whatever the bitmap makes of it (marked everywhere, or an
    overflow that invalidates it), the outputs stay equal; the renderer's own choice then switches the tiers off."""
    from arah_release_amd import config
    dev = torch.device("cuda:0")
    model, _ = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
    with torch.no_grad():
        model.deviation_decoder.variance.fill_(3e-2)
    inputs = scene.make_inputs(128, 128, frame_idx=1, device=dev)
    res = _both_ways(model, inputs, 64)
    c = _assert_same(res, "beta 3e-2")
    assert c["n_tier_samples_skipped"] < 0.5 * (c["n_tier_samples_p1"] + c["n_tier_samples_p2"])
    # the product's choice: after a tiered frame whose skipped share is small, the following frames run untiered
    idhr = model.idhr_network
    idhr.adaptive_shading, idhr.tiering = True, True
    tracer = idhr.ray_tracer
    with torch.no_grad():
        for _ in range(3):
            model(dict(inputs), eval=True)
            torch.cuda.synchronize()
    assert idhr._tier_off or idhr._shade_full


@gpu
def test_tiers_batch_of_two_views_and_empty_rays(scene):
    """Two cameras in one call (rays_per_cam < n), and rays whose interval is empty (near == far)."""
    from arah_release_amd import config
    dev = torch.device("cuda:0")
    model, _ = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
    a = scene.make_inputs(96, 96, frame_idx=2, max_rays=1500, device=dev)
    two = dict(a)
    for k in ("ray_dirs", "body_bounds_intersections", "cam_loc", "pose", "body_mask"):
        if k in a and torch.is_tensor(a[k]):
            two[k] = torch.cat([a[k], a[k]], dim=0)
    two["cam_loc"] = torch.cat([a["cam_loc"], a["cam_loc"] + torch.tensor([[0.05, 0.0, 0.0]], device=dev)], dim=0)
    nf = two["body_bounds_intersections"].clone()
    nf[0, :40, 0] = nf[0, :40, 1]                # empty intervals (near == far: legal, RT:182 asserts near <= far)
    two["body_bounds_intersections"] = nf
    idhr = model.idhr_network
    outs = {}
    for on in (False, True):
        idhr.tiering, idhr.adaptive_shading = on, False
        with torch.no_grad():
            outs[on] = model(dict(two), eval=True)
    idhr.tiering, idhr.adaptive_shading = True, True
    for key in ("rgb_values", "network_body_mask", "points_cam"):
        assert torch.equal(outs[False][key], outs[True][key]), key


@gpu
def test_inference_caches_live_outside_the_modules(scene):
    """The inference caches (folded weight-norm layers with their event, the captured hypernetwork graph) are keyed weakly by
    the modules, not stored in them: nothing un-picklable (torch.cuda.Event, CUDAGraph) sits in any module's __dict__ after a
    frame (round 5's advisor: deepcopy / pickle of such a model raised on them; what still stands in the way of deepcopy is
    torch's own weight_norm, as in the reference).  A write through .data is invisible to the caches' version keys:
    renderer.invalidate_caches drops them, and load_state_dict does so by itself."""
    import pickle
    from arah_release_amd import config, renderer
    dev = torch.device("cuda:0")
    model, _ = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
    inputs = scene.make_inputs(64, 64, frame_idx=0, device=dev)
    with torch.no_grad():
        a = model(dict(inputs), eval=True)["rgb_values"].clone()
        assert len(renderer._FOLDED) >= 2 and len(renderer._GRAPHED) >= 1          # the caches are in use
        for m in model.modules():
            for k, v in m.__dict__.items():
                assert not isinstance(v, (torch.cuda.Event, torch.cuda.CUDAGraph)), (type(m).__name__, k)
                assert not k.startswith(("_arah_folded", "_graphed", "_train_graphed")), (type(m).__name__, k)
        pickle.loads(pickle.dumps(model.deviation_decoder))
        lin = model.color_decoder.lin5
        lin.weight_g.data.mul_(0.5)                     # a write the version counter does not see ...
        renderer.invalidate_caches(model)               # ... announced
        c = model(dict(inputs), eval=True)["rgb_values"].clone()
        assert not torch.equal(a, c)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        for k in sd:   # the colour MLP is registered under two names (model.color_decoder, idhr_network.rendering_network)
            if k.endswith("lin5.weight_g"):
                sd[k] = sd[k] * 2.0
        model.load_state_dict(sd)                       # writes through copy_ AND drops the caches (post-hook)
        d = model(dict(inputs), eval=True)["rgb_values"]
        assert torch.equal(a, d)
