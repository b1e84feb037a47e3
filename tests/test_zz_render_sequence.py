"""Sequence renderer (several frames in flight on several HIP streams).  Kept in its own file so that it runs LAST: the
HIP runtime of this image has stalled in multi-stream runs, and a stalled launch cannot be
interrupted from Python -- the per-test timeout then ends the whole pytest process."""
import pytest
import torch

gpu = pytest.mark.gpu


@gpu
@pytest.mark.timeout(150)
@pytest.mark.parametrize("n_streams", [2, 3, None])
def test_render_sequence_matches_frame_by_frame(scene, n_streams):
    """renderer.render_sequence keeps several frames in flight (frame k on stream k mod n, own scratch): every frame's
    outputs are bit-identical to rendering the frames one after the other, in order, for ragged frame sizes."""
    from arah_release_amd import config, renderer
    dev = torch.device("cuda:0")
    model, cfg = config.build_synthetic_model("zju377_mono", device=dev)
    frames = [scene.make_inputs(s, s, frame_idx=f, device=dev) for f, s in ((0, 256), (3, 192), (7, 256), (11, 128), (5, 256))]
    with torch.no_grad():
        ref = [model(dict(f), eval=True) for f in frames]
    got = renderer.render_sequence(model, [dict(f) for f in frames], n_streams=n_streams, eval=True)
    assert len(got) == len(ref)
    for a, b in zip(ref, got):
        for k in ("rgb_values", "network_body_mask", "points_cam"):
            assert torch.equal(a[k], b[k]), k
    # twice in a row (streams and scratch are reused), and the degenerate cases
    again = renderer.render_sequence(model, [dict(f) for f in frames], n_streams=n_streams, eval=True)
    assert all(torch.equal(a["rgb_values"], b["rgb_values"]) for a, b in zip(ref, again))
    assert renderer.render_sequence(model, [], n_streams=n_streams) == []
    one = renderer.render_sequence(model, [dict(frames[0])], n_streams=1, eval=True)
    assert torch.equal(one[0]["rgb_values"], ref[0]["rgb_values"])
    assert len(model.idhr_network.ray_tracer.workspaces()) >= (n_streams or renderer.frames_in_flight(len(frames)))


@gpu
@pytest.mark.timeout(200)
def test_mesh_branch_in_flight_matches_frame_by_frame(scene):
    """renderer.map_in_flight with the test.py frame (render + canonical mesh + three normal maps, what test_sequence.render
    sends through it): bit-identical to one frame after the other."""
    from arah_release_amd import config, renderer
    dev = torch.device("cuda:0")
    model, cfg = config.build_synthetic_model("zju377_mono", device=dev)
    frames = [scene.make_inputs(128, 128, frame_idx=f, device=dev) for f in (0, 4, 9, 2, 6)]
    fn = lambda f: model(dict(f), gen_cano_mesh=True, eval=True)   # noqa: E731
    with torch.no_grad():
        ref = [fn(f) for f in frames]
    got = renderer.map_in_flight(fn, frames, n_streams=3, owner=model)
    keys = [k for k in ref[0] if torch.is_tensor(ref[0][k])]
    assert {"rgb_values", "output_normal", "normal_cano_front", "normal_cano_back"} <= set(keys)
    for a, b in zip(ref, got):
        for k in keys:
            assert torch.equal(a[k], b[k]), k
