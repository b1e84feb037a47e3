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
