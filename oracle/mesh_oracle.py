"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the rasterisation / normal-map step of the gen_cano_mesh branch
(reference metaavatar_render/models/__init__.py:203-311).  Never imported by the product.

The reference delegates this step to pytorch3d 0.6.1 (``MeshRasterizer`` with one face per pixel, ``pix_to_face``),
which is neither in its tree nor in this image: PARITY UNPINNED for the rasteriser itself.  What this file pins is
the build's own arithmetic -- the same documented semantics (pixel (i, j) takes the nearest face whose projection
covers the pixel centre (j + 0.5, i + 0.5); faces reaching the near plane are dropped) written as plain numpy loops.
"""
import numpy as np


def rasterize_np(tri_uvz, H, W, z_near=1e-4):
    """tri_uvz (F,3,3) float32 (u, v, depth) -> pix_to_face (H,W) int64 (-1 = background); ties go to the lower face
    index at equal depth bits, like a min over (depth, face) keys."""
    tri = np.asarray(tri_uvz, np.float32)
    best_z = np.full((H, W), np.inf, np.float32)
    best_f = -np.ones((H, W), np.int64)
    for f in range(tri.shape[0]):
        (x0, y0, z0), (x1, y1, z1), (x2, y2, z2) = tri[f]
        if not (z0 > z_near and z1 > z_near and z2 > z_near):
            continue
        area = np.float32((x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0))
        if area == 0 or not np.isfinite(area):
            continue
        inv = np.float32(1.0) / area
        j0 = max(0, int(np.floor(min(x0, x1, x2) - 0.5)))
        j1 = min(W - 1, int(np.ceil(max(x0, x1, x2) - 0.5)))
        i0 = max(0, int(np.floor(min(y0, y1, y2) - 0.5)))
        i1 = min(H - 1, int(np.ceil(max(y0, y1, y2) - 0.5)))
        for i in range(i0, i1 + 1):
            for j in range(j0, j1 + 1):
                px, py = np.float32(j + 0.5), np.float32(i + 0.5)
                w0 = ((x1 - px) * (y2 - py) - (x2 - px) * (y1 - py)) * inv
                w1 = ((x2 - px) * (y0 - py) - (x0 - px) * (y2 - py)) * inv
                w2 = np.float32(1.0) - w0 - w1
                if w0 < 0 or w1 < 0 or w2 < 0:
                    continue
                z = w0 * z0 + w1 * z1 + w2 * z2
                if z < best_z[i, j] or (z == best_z[i, j] and f < best_f[i, j]):
                    best_z[i, j] = z
                    best_f[i, j] = f
    return best_f


# ------------------------------------------------------------------------------------------------------------------
# mesh queries of the training data path (reference im2mesh/data/zju_mocap.py:461-543)
# ------------------------------------------------------------------------------------------------------------------
def check_mesh_contains_np(verts, faces, points, resolution=512, chunk=256):
    """im2mesh/utils/libmesh/inside_mesh.py:4-160 without the Cython triangle hash (which only pre-selects candidate
    triangles; every candidate goes through the same strict 2-D containment test): all (point, triangle) pairs, double
    precision, one rounding per operation.  PINNED by fixture F10 (the reference's own module, run with a brute-force
    stand-in for the hash)."""
    tri = np.asarray(verts, np.float64)[np.asarray(faces)]                       # :13
    lo = tri.reshape(-1, 3).min(0)
    hi = tri.reshape(-1, 3).max(0)
    scale = (resolution - 1) / (hi - lo)                                         # :21
    translate = 0.5 - scale * lo
    tri = scale * tri + translate
    pts = scale * np.asarray(points, np.float64) + translate
    inside_aabb = np.all((0 <= pts) & (pts <= resolution), axis=1)               # :43-44
    t1, t2, t3 = tri[:, 0], tri[:, 1], tri[:, 2]
    A00, A01 = t1[:, 0] - t3[:, 0], t2[:, 0] - t3[:, 0]                          # :138-139 (transposed)
    A10, A11 = t1[:, 1] - t3[:, 1], t2[:, 1] - t3[:, 1]
    detA = A00 * A11 - A01 * A10
    ok_tri = np.abs(detA) != 0.0
    s_det, a_det = np.sign(detA), np.abs(detA)
    v1, v2 = t3 - t1, t2 - t1                                                    # :105-106
    nrm = np.cross(v1, v2)
    n2 = nrm[:, 2]
    s_n2, a_n2 = np.sign(n2), np.abs(n2)
    out = np.zeros(len(pts), bool)
    for c0 in range(0, len(pts), chunk):
        p = pts[c0:c0 + chunk]
        y0 = p[:, None, 0] - t3[None, :, 0]
        y1 = p[:, None, 1] - t3[None, :, 1]
        u = (A11[None] * y0 - A01[None] * y1) * s_det[None]                      # :152-153
        v = (-A10[None] * y0 + A00[None] * y1) * s_det[None]
        suv = u + v
        hit = ok_tri[None] & (0 < u) & (u < a_det[None]) & (0 < v) & (v < a_det[None]) & (0 < suv) & (suv < a_det[None])
        alpha = nrm[None, :, 0] * (t1[None, :, 0] - p[:, None, 0]) + nrm[None, :, 1] * (t1[None, :, 1] - p[:, None, 1])
        depth = t1[None, :, 2] * a_n2[None] + alpha * s_n2[None]                 # :121-122
        valid = hit & (a_n2[None] != 0)
        rhs = p[:, None, 2] * a_n2[None]
        n0 = (valid & (depth >= rhs)).sum(1)                                     # :64-69
        n1 = (valid & (depth < rhs)).sum(1)
        out[c0:c0 + chunk] = (n0 % 2 == 1) & (n1 % 2 == 1)
    return out & inside_aabb


def point_mesh_np(verts, faces, points, chunk=128):
    """igl.point_mesh_squared_distance + igl.barycentric_coordinates_tri (libigl is neither in the reference tree nor in
    this image: PARITY UNPINNED against libigl, the quantities are geometrically determined).  Written differently from
    the kernel on purpose: closest point = the orthogonal projection when it falls inside the triangle, else the nearest
    of the three clamped edge projections.  -> d2 (P,), face (P,) lowest index among exact ties, closest (P,3), bary (P,3)"""
    V = np.asarray(verts, np.float64)
    F = np.asarray(faces)
    P = np.asarray(points, np.float64)
    a, b, c = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    n = np.cross(b - a, c - a)
    nn = (n * n).sum(1)
    d2o = np.empty(len(P)); fo = np.empty(len(P), np.int64); co = np.empty((len(P), 3)); bo = np.empty((len(P), 3))

    def seg(p, s, e):
        d = e - s
        t = np.clip(((p - s[None]) * d[None]).sum(-1) / np.maximum((d * d).sum(-1), 1e-300)[None], 0.0, 1.0)
        return s[None] + t[..., None] * d[None]

    for c0 in range(0, len(P), chunk):
        p = P[c0:c0 + chunk, None, :]                                            # (p,1,3)
        # plane projection and its barycentric coordinates
        t = ((p - a[None]) * n[None]).sum(-1) / np.maximum(nn, 1e-300)[None]
        q = p - t[..., None] * n[None]
        wa = (np.cross(b[None] - q, c[None] - q) * n[None]).sum(-1) / np.maximum(nn, 1e-300)[None]
        wb = (np.cross(c[None] - q, a[None] - q) * n[None]).sum(-1) / np.maximum(nn, 1e-300)[None]
        wc = 1.0 - wa - wb
        inside = (wa >= 0) & (wb >= 0) & (wc >= 0) & (nn[None] > 0)
        cands = [q, seg(p, a, b), seg(p, b, c), seg(p, c, a)]
        d2s = [np.where(inside, ((p - q) ** 2).sum(-1), np.inf)] + [((p - s) ** 2).sum(-1) for s in cands[1:]]
        d2s = np.stack(d2s, 0)
        k = d2s.argmin(0)
        cp = np.take_along_axis(np.stack(cands, 0), k[None, ..., None], 0)[0]
        d2 = np.take_along_axis(d2s, k[None], 0)[0]
        f = d2.argmin(1)
        idx = np.arange(len(f))
        d2o[c0:c0 + chunk] = d2[idx, f]
        fo[c0:c0 + chunk] = f
        cpt = cp[idx, f]
        co[c0:c0 + chunk] = cpt
        # barycentric weights of the closest point in its face (areas)
        A, B, C, N = a[f], b[f], c[f], n[f]
        den = np.maximum((N * N).sum(1), 1e-300)
        ba = (np.cross(B - cpt, C - cpt) * N).sum(1) / den
        bb = (np.cross(C - cpt, A - cpt) * N).sum(1) / den
        bo[c0:c0 + chunk] = np.stack([ba, bb, 1.0 - ba - bb], 1)
    return d2o, fo, co, bo
