"""Mesh queries and samplers of the training data path (reference im2mesh/data/zju_mocap.py:455-543).
Containment is pinned by fixture F10 (the reference's own libmesh module); closest point / barycentric weights by the
numpy oracle (libigl is not available anywhere here)."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import mesh_oracle

gpu = pytest.mark.gpu


def test_oracle_containment_against_reference():
    g = golden("f10_mesh_contains.npz")
    got = mesh_oracle.check_mesh_contains_np(g["verts"], g["faces"], g["points"])
    np.testing.assert_array_equal(got, g["contains"])
    assert 500 < int(g["contains"].sum()) < 3000
    # the 64 mesh vertices used as query points are the degenerate case (rays through vertices and edges): the
    # reference answers "not contained" for every one of them
    assert not g["contains"][-64:].any()


def test_oracle_closest_point_on_a_cube():
    V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], np.float32)
    F = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6],
                  [3, 0, 4], [3, 4, 7]])
    P = np.random.RandomState(0).rand(500, 3) * 1.6 - 0.3
    inside = np.all((P > 0) & (P < 1), 1)
    np.testing.assert_array_equal(mesh_oracle.check_mesh_contains_np(V, F, P), inside)
    d2, f, cp, b = mesh_oracle.point_mesh_np(V, F, P)
    q = np.clip(P, 0, 1)
    truth = np.where(inside, np.minimum(P, 1 - P).min(1) ** 2, ((P - q) ** 2).sum(1))
    np.testing.assert_allclose(d2, truth, atol=1e-14)
    np.testing.assert_allclose(b.sum(1), 1.0, atol=1e-12)
    tri = V[F[f]].astype(np.float64)
    np.testing.assert_allclose((tri * b[..., None]).sum(1), cp, atol=1e-12)


def test_sample_surface_is_area_proportional_and_on_the_mesh():
    from arah_release_amd import data
    g = golden("f10_mesh_contains.npz")
    v, f = torch.from_numpy(g["verts"]), torch.from_numpy(g["faces"])
    gen = torch.Generator().manual_seed(3)
    pts, fi = data.sample_surface(v, f, 20000, gen)
    tri = v[f.long()]
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    area = torch.linalg.cross(e1, e2).norm(dim=-1) * 0.5
    # every point lies in the plane of its face, inside the triangle
    n = torch.nn.functional.normalize(torch.linalg.cross(e1, e2), dim=-1)
    off = ((pts - tri[fi, 0]) * n[fi]).sum(-1).abs()
    assert float(off.max()) < 1e-5
    d2, _, _, bary = mesh_oracle.point_mesh_np(g["verts"], g["faces"], pts[:500].numpy())
    assert d2.max() < 1e-10 and bary.min() > -1e-5
    # area-proportional: compare the hit histogram of the faces, binned by area rank, with the area shares
    order = torch.argsort(area)
    bins = torch.chunk(order, 8)
    hits = torch.bincount(fi, minlength=f.shape[0]).float()
    for b in bins:
        share, got = float(area[b].sum() / area.sum()), float(hits[b].sum() / hits.sum())
        assert abs(got - share) < 0.02, (share, got)


@gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_mesh_query_against_reference_and_oracle(dtype):
    from arah_release_amd import hip
    g = golden("f10_mesh_contains.npz")
    dev = torch.device("cuda:0")
    v, f = torch.from_numpy(g["verts"]).to(dev), torch.from_numpy(g["faces"]).to(dev)
    pts_np = g["points"].astype(np.float32 if dtype == torch.float32 else np.float64)
    d2, face, closest, bary, inside = hip.mesh_query(v, f, torch.from_numpy(pts_np).to(dev))
    if dtype == torch.float64:    # the reference's own answer for exactly these float64 points, degenerate ones included
        np.testing.assert_array_equal(inside.cpu().numpy(), g["contains"])
    ref_in = mesh_oracle.check_mesh_contains_np(g["verts"], g["faces"], pts_np.astype(np.float64))
    np.testing.assert_array_equal(inside.cpu().numpy(), ref_in)
    rd2, rf, rcp, rb = mesh_oracle.point_mesh_np(g["verts"], g["faces"], pts_np.astype(np.float64))
    np.testing.assert_allclose(d2.cpu().numpy(), rd2, rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(closest.cpu().numpy(), rcp, atol=1e-7)
    same = face.cpu().numpy() == rf                      # ties (closest point on a shared edge / vertex) may name a neighbour
    assert same.mean() > 0.9
    np.testing.assert_allclose(bary.cpu().numpy()[same], rb[same], atol=1e-6)
    tri = g["verts"][g["faces"][face.cpu().numpy()]].astype(np.float64)
    np.testing.assert_allclose((tri * bary.cpu().numpy()[..., None]).sum(1), closest.cpu().numpy(), atol=1e-12)
    # the lowest face index wins exact ties: a query AT a vertex is at distance 0 from every incident face
    d2v, fv, _, _, _ = hip.mesh_query(v, f, v[:200].to(dtype))
    faces_np = g["faces"]
    first = np.array([np.nonzero((faces_np == i).any(1))[0][0] for i in range(200)])
    assert float(d2v.max()) == 0.0
    np.testing.assert_array_equal(fv.cpu().numpy(), first)
    # empty query set
    out = hip.mesh_query(v, f, torch.zeros(0, 3, dtype=dtype, device=dev))
    assert all(o.shape[0] == 0 for o in out)


@gpu
def test_training_samples_properties():
    """zju_mocap.py:455-543 on the device: shapes, and every published property of the three point sets."""
    from arah_release_amd import data
    g = golden("f10_mesh_contains.npz")
    dev = torch.device("cuda:0")
    v, f = torch.from_numpy(g["verts"]).to(dev), torch.from_numpy(g["faces"]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(11)
    # skinning weights: soft assignment of every vertex to 24 "parts" stacked along y (hands = parts 22, 23 at the top)
    yc = torch.linspace(-0.95, 0.75, 24, device=dev)
    w = torch.softmax(-((v[:, 1:2] - yc[None]) ** 2) / 0.003, dim=-1)
    center = v.mean(0)
    cmin, cmax = (v - center).min(), (v - center).max()
    span = cmax - cmin
    unnorm = lambda p: (p / 2.0 + 0.5) * 1.1 * span + cmin - span * 0.05 + center
    for reg, inside in ((True, True), (False, False)):
        out = data.training_samples(v, f, w, cmin, cmax, center, sample_reg_surface=reg, sample_inside=inside,
                                    off_surface_thr=0.01, inside_thr=1e-4, generator=gen)
        pu = out["points_uniform"]
        assert tuple(pu.shape) == (1024, 3) and float(pu.abs().max()) <= 1.0
        d2, _, _, _ = mesh_oracle.point_mesh_np(g["verts"], g["faces"], unnorm(pu).double().cpu().numpy())
        assert (d2 > 0.01).all()
        assert not mesh_oracle.check_mesh_contains_np(g["verts"], g["faces"], unnorm(pu).double().cpu().numpy()).any()
        sw = out["sampled_weights"]
        np.testing.assert_allclose(sw.sum(-1).cpu().numpy(), 1.0, atol=1e-5)
        if reg:
            ps = out["points_skinning"]
            assert tuple(ps.shape) == (1024, 3) and tuple(sw.shape) == (1024, 24)
            d2s, fs, _, bs = mesh_oracle.point_mesh_np(g["verts"], g["faces"], ps.double().cpu().numpy())
            assert d2s.max() < 1e-10
            ref_w = (w.cpu().numpy()[g["faces"][fs]] * bs[..., None]).sum(1)
            np.testing.assert_allclose(sw.cpu().numpy(), ref_w, atol=2e-4)
        else:
            assert tuple(out["points_skinning"].shape) == (24, 3) and torch.equal(sw, torch.eye(24, device=dev))
        if inside:
            pin = unnorm(out["points_inside"])
            assert tuple(pin.shape) == (1024, 3)
            inn = mesh_oracle.check_mesh_contains_np(g["verts"], g["faces"], pin.double().cpu().numpy())
            assert inn.mean() > 0.9            # all but (possibly) the 22 part centroids, which are appended unconditionally
        else:
            assert "points_inside" not in out
    with pytest.raises(ValueError):
        data.training_samples(v, f, w, cmin, cmax, center, off_surface_thr=1e9, generator=gen)
