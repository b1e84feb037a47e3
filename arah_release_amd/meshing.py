"""Canonical-mesh branch of the model entry (``gen_cano_mesh=True``, reference metaavatar_render/models/__init__.py:
203-311 and utils/sdf_meshing.py:13-114), on the device end to end:

  1. SDF on the 256^3 lattice of [-1,1]^3 -- one launch of the SDF kernel (``arah_sdf_grid``), the values never
     leave HBM (the reference: 64 chunks with a ``.cpu()`` copy each, sdf_meshing.py:44-57);
  2. marching cubes at level 0 as three kernels (``arah_marching_cubes``, csrc/mcubes.hpp: count per lattice row, scan,
     emit) whose triangle COUNT stays on the device -- no host round trip anywhere in the branch, so that the frames of a
     test sequence keep overlapping (``marching_cubes`` below is the same extraction as tensor operations: the
     specification the kernel is tested against, and what runs for volumes that live on the host);
  3. forward skinning of the vertices (``arah_skin_lbs``), projection, rasterisation (``arah_rasterize``) and the three
     normal maps ``output_normal`` / ``normal_cano_front`` / ``normal_cano_back`` (1,512,512,3).

Third-party pieces of the reference that are not in its tree and absent from this image, restated from their
documented behaviour (parity unpinned at triangle level):
  * ``skimage.measure.marching_cubes_lewiner`` (scikit-image 0.18): same level set, same linear interpolation of the
    crossing points along lattice edges, triangles oriented like skimage's default ``gradient_direction='descent'``
    (right-hand normals point towards DECREASING values).  The triangulation inside a cell comes from a case table
    generated here (face-consistent loops, fan triangulation) instead of Lewiner's 33-case tables: the surface is the
    same to within the cell, individual facets differ.
  * ``pytorch3d`` 0.6.1 ``cameras_from_opencv_projection``, ``look_at_view_transform``, ``FoVPerspectiveCameras``
    (fov 60 deg) and ``MeshRasterizer`` (``pix_to_face``, one face per pixel, no blur, no culling): pixel (i, j) takes
    the nearest face covering its centre.
"""
import math
import os

import numpy as np
import torch

# cube corners (dx, dy, dz), the 12 edges between them and the 6 faces as cyclic corner quadruples
CORNERS = ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1))
EDGES = ((0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7))
FACES = ((0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7))
_EDGE_ID = {frozenset(e): i for i, e in enumerate(EDGES)}
_TABLE = None


def case_table():
    """tri_edges (256, 3*T) int8 (edge ids, -1 padded) and n_tri (256,) for every inside/outside pattern of the 8
    corners (bit c set = corner c inside).  Per face the crossed edges are joined into segments -- with four crossings
    the segments cut off the INSIDE corners, a rule that depends on the face's corner signs only, so neighbouring cells
    agree on their shared face and the mesh has no cracks --, segments chain into closed loops, loops are fanned."""
    global _TABLE
    if _TABLE is not None:
        return _TABLE
    tris = []
    for case in range(256):
        inside = [(case >> c) & 1 for c in range(8)]
        nbr = {}
        for face in FACES:
            crossed = []
            for i in range(4):
                a, b = face[i], face[(i + 1) % 4]
                if inside[a] != inside[b]:
                    crossed.append(_EDGE_ID[frozenset((a, b))])
            if len(crossed) == 2:
                segs = [tuple(crossed)]
            elif len(crossed) == 4:
                segs = []
                for i in range(4):
                    if inside[face[i]]:
                        segs.append((_EDGE_ID[frozenset((face[i - 1], face[i]))],
                                     _EDGE_ID[frozenset((face[i], face[(i + 1) % 4]))]))
            else:
                segs = []
            for a, b in segs:
                nbr.setdefault(a, []).append(b)
                nbr.setdefault(b, []).append(a)
        assert all(len(v) == 2 for v in nbr.values()), case
        out, seen = [], set()
        for start in sorted(nbr):
            if start in seen:
                continue
            loop, prev, cur = [start], None, start
            while True:
                a, b = nbr[cur]
                nxt = a if a != prev else b
                if nxt == start:
                    break
                loop.append(nxt)
                prev, cur = cur, nxt
            seen.update(loop)
            assert len(loop) >= 3, case
            for i in range(1, len(loop) - 1):
                out += [loop[0], loop[i], loop[i + 1]]
        tris.append(out)
    width = max(len(t) for t in tris)
    table = -np.ones((256, width), np.int8)
    for c, t in enumerate(tris):
        table[c, :len(t)] = t
    _TABLE = (table, np.array([len(t) // 3 for t in tris], np.int64))
    return _TABLE


def marching_cubes(sdf, level=0.0):
    """sdf (N,N,N) tensor indexed [ix,iy,iz] on the lattice of [-1,1]^3 -> triangle soup (F,3,3) of coordinates in
    [-1,1]^3 (sdf_meshing.py:83-101: vertex = origin + index * voxel_size), right-hand normals towards decreasing
    values.  Runs on the tensor's device."""
    dev = sdf.device
    N = sdf.shape[0]
    vs = 2.0 / (N - 1)
    table_np, ntri_np = case_table()
    table = torch.from_numpy(table_np.astype(np.int64)).to(dev)
    ntri = torch.from_numpy(ntri_np).to(dev)
    corners = torch.tensor(CORNERS, device=dev)
    edges = torch.tensor(EDGES, device=dev)
    inside = sdf < level
    case = torch.zeros(N - 1, N - 1, N - 1, dtype=torch.int64, device=dev)
    for c, (dx, dy, dz) in enumerate(CORNERS):
        case += inside[dx:N - 1 + dx, dy:N - 1 + dy, dz:N - 1 + dz].to(torch.int64) << c
    cells = torch.nonzero((case != 0) & (case != 255))                      # (M,3)
    if cells.shape[0] == 0:
        return torch.zeros(0, 3, 3, device=dev)
    ccase = case[cells[:, 0], cells[:, 1], cells[:, 2]]
    cnt = ntri[ccase]
    owner = torch.repeat_interleave(torch.arange(cells.shape[0], device=dev), cnt)           # cell of every triangle
    first = torch.cumsum(cnt, 0) - cnt
    slot = torch.arange(owner.shape[0], device=dev) - first[owner]                            # its index in the cell
    e = table[ccase[owner].unsqueeze(1), (slot * 3).unsqueeze(1) + torch.arange(3, device=dev)]   # (F,3) edge ids
    base = cells[owner]                                                                        # (F,3)
    pa = base.unsqueeze(1) + corners[edges[e][..., 0]]                                        # (F,3,3) lattice indices
    pb = base.unsqueeze(1) + corners[edges[e][..., 1]]
    # every lattice edge is interpolated from its lower to its higher end, whichever cell asks: shared vertices come out
    # bit-identical on both sides
    swap = (pa > pb).any(-1, keepdim=True)
    pa, pb = torch.where(swap, pb, pa), torch.where(swap, pa, pb)
    va = sdf[pa[..., 0], pa[..., 1], pa[..., 2]] - level
    vb = sdf[pb[..., 0], pb[..., 1], pb[..., 2]] - level
    t = (va / (va - vb)).clamp(0.0, 1.0).unsqueeze(-1)
    verts = (pa.float() + t * (pb - pa).float()) * vs - 1.0
    # orientation: the cell's corner values give the gradient direction; normals must point DOWN the gradient
    cv = torch.stack([sdf[base[:, 0] + dx, base[:, 1] + dy, base[:, 2] + dz] for dx, dy, dz in CORNERS], dim=1)   # (F,8)
    cf = corners.float()
    grad = torch.stack([(cv * (2 * cf[:, k] - 1)).sum(1) for k in range(3)], dim=1)
    nrm = torch.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0], dim=1)
    flip = (nrm * grad).sum(1) > 0
    verts = torch.where(flip[:, None, None], verts[:, [0, 2, 1]], verts)
    return verts


def face_normals(tri):
    """Unit right-hand normals of a triangle soup (F,3,3) (pytorch3d Meshes.faces_normals_packed)."""
    n = torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=1)
    return n / n.norm(dim=1, keepdim=True).clamp_min(1e-20)


def project_opencv(pts, cam_rot, cam_trans, K):
    """World points (...,3) -> (u, v, depth) with x_cam = R x + t, u = fx X/Z + cx (cameras_from_opencv_projection)."""
    xc = pts @ cam_rot.t() + cam_trans
    z = xc[..., 2]
    u = K[0, 0] * xc[..., 0] / z + K[0, 2]
    v = K[1, 1] * xc[..., 1] / z + K[1, 2]
    return torch.stack([u, v, z], dim=-1)


_LOOKAT = {}   # (device, azimuth, distance) -> (camera position (3,), view axes as columns (3,3)) on the device


def _lookat(device, azim_deg, dist):
    """look_at_view_transform(dist, 0, azim): the camera position and the matrix whose columns are the view axes.  Built once
    per device and view: a tensor made from Python numbers is a blocking host -> device copy, which waits for everything the
    stream has queued -- inside a frame that would be the whole render in front of the mesh branch."""
    key = (device, float(azim_deg), float(dist))
    if key not in _LOOKAT:
        a = math.radians(azim_deg)
        cam = torch.tensor([dist * math.sin(a), 0.0, dist * math.cos(a)])
        z_axis = -cam / cam.norm()
        up = torch.tensor([0.0, 1.0, 0.0])
        x_axis = torch.cross(up, z_axis, dim=0)
        x_axis = x_axis / x_axis.norm()
        y_axis = torch.cross(z_axis, x_axis, dim=0)
        R = torch.stack([x_axis, y_axis, z_axis], dim=1)          # columns = view axes
        _LOOKAT[key] = (cam.to(device), R.to(device))
    return _LOOKAT[key]


def project_lookat(pts, azim_deg, size, dist=2.0, fov_deg=60.0):
    """look_at_view_transform(dist, elev 0, azim) + FoVPerspectiveCameras(fov 60): canonical points -> (u, v, depth).
    pytorch3d's view space has +X left, +Y up, +Z into the screen; NDC (1,1) is the top-left pixel corner."""
    cam, R = _lookat(pts.device, azim_deg, dist)
    xv = (pts - cam) @ R
    f = 1.0 / math.tan(math.radians(fov_deg) / 2.0)
    z = xv[..., 2]
    u = (1.0 - f * xv[..., 0] / z) * size / 2.0
    v = (1.0 - f * xv[..., 1] / z) * size / 2.0
    return torch.stack([u, v, z], dim=-1)


def normal_image(pix_to_face, normals, background):
    """normals (F,3) gathered per pixel, `background` elsewhere, mapped to [0,1] like models/__init__.py:247,278.
    (A gather and a select: boolean-mask indexing would cost a device -> host round trip for the number of pixels.)"""
    fg = (pix_to_face >= 0).unsqueeze(-1)
    img = torch.where(fg, normals[pix_to_face.clamp_min(0)], torch.full((), float(background), device=normals.device))
    return ((img + 1.0) / 2.0).clip(0.0, 1.0).unsqueeze(0)


# Triangle capacity of the device-side extraction per device, and the counts of earlier calls on their way to the host
# (asynchronous copies into pinned memory: nothing here ever waits for the GPU).  A count that turns out to have exceeded
# the capacity raises it for the calls that follow and is reported: that call's mesh was truncated.
_MC_STATE = {}
MC_DEFAULT_CAP = 1 << 20


def _mc_state(dev):
    st = _MC_STATE.get(dev)
    if st is None:
        st = _MC_STATE[dev] = {"cap": MC_DEFAULT_CAP, "pending": [], "free": [], "overflowed": 0, "last_count": None}
    return st


def _mc_poll(st, wait=False):
    import warnings
    keep = []
    for ev, host, cap in st["pending"]:
        if wait:
            ev.synchronize()
        if ev.query():
            n = int(host[0])
            st["last_count"] = n
            st["free"].append((ev, host))
            if n <= cap and not st["overflowed"]:
                # every frame pushes the whole buffer through un-normalisation, skinning, projection and three rasterisations:
                # size it for the level set this subject actually has (twice the last count, a power of two, never above the
                # default; one overflow pins the raised capacity for good)
                st["cap"] = min(MC_DEFAULT_CAP, max(1 << 16, 1 << int(math.ceil(math.log2(2.0 * max(n, 1))))))
            if n > cap:
                st["overflowed"] += 1
                st["cap"] = max(st["cap"], 1 << int(math.ceil(math.log2(1.5 * n))))
                warnings.warn("canonical mesh: the level set has %d triangles, the device buffer held %d -- that frame's mesh was "
                              "truncated; capacity raised to %d for the following frames" % (n, cap, st["cap"]))
        else:
            keep.append((ev, host, cap))
    st["pending"] = keep


def mesh_counts(device, wait=True):
    """(triangles of the last finished extraction on `device`, number of truncated extractions so far); wait=True drains the
    outstanding count copies first (tests, end of a sequence)."""
    st = _mc_state(torch.device(device))
    _mc_poll(st, wait=wait)
    return st["last_count"], st["overflowed"]


def canonical_mesh_outputs(frame, ws, inputs, rasterize_fn=None, n_side=256, image_size=512, tri=None, want_tri=True):
    """The three normal maps of the gen_cano_mesh branch + the canonical triangle soup (normalised coordinates).
    frame: packed hip.Frame of the current pose; inputs: the model's input dict (coord_min/max, center, trans,
    cam_rot, cam_trans, intrinsics).  tri: a triangle soup to use instead of meshing the SDF (fixture F18 injects the mesh
    the reference's own branch was run on).  want_tri=False (the model entry): the soup is not returned and the call makes
    no device -> host round trip at all -- the mesh lives in a fixed-capacity buffer whose tail is degenerate triangles, its
    size stays on the device (hip.marching_cubes, hip.skin_lbs_counted); want_tri=True trims the soup to its size, which
    waits for the GPU."""
    from . import hip, training
    rasterize_fn = rasterize_fn or hip.rasterize
    with torch.no_grad():
        n_dev = None
        if tri is None:
            # the lattice only where the level set can pass (csrc/tier.hpp: same triangles as the full lattice, ~6 % of its
            # 16.8 M evaluations); ARAH_MESH_BAND=0: every lattice point, like sdf_meshing.py:44-57
            if n_side >= 33 and os.environ.get("ARAH_MESH_BAND", "1") != "0":
                sdf, _ = hip.sdf_grid_band(frame, ws, n_side)
            else:
                sdf = hip.sdf_grid(frame, ws, n_side)
            st = _mc_state(sdf.device)
            _mc_poll(st)
            tri, n_dev = hip.marching_cubes(sdf, 0.0, st["cap"])                         # (cap,3,3) in [-1,1]^3, zero tail
            ev, host = st["free"].pop() if st["free"] else (torch.cuda.Event(), torch.empty(1, dtype=torch.int32).pin_memory())
            host.copy_(n_dev, non_blocking=True)
            ev.record()
            st["pending"].append((ev, host, st["cap"]))
        F = tri.shape[0]
        cmin, cmax, center = inputs["coord_min"][:1], inputs["coord_max"][:1], inputs["center"][:1]
        x_hat = training.unnormalize_canonical_points(tri.reshape(1, -1, 3), cmin, cmax, center)[0]
        if n_dev is None:
            _, x_bar, _ = hip.skin_lbs(frame, ws, x_hat)
        else:
            x_bar = hip.skin_lbs_counted(frame, ws, x_hat, n_dev, per_item=3)            # zero beyond the mesh: degenerate
        posed = (x_bar + inputs["trans"].reshape(1, 3)).reshape(F, 3, 3)
        cam_rot, cam_trans, K = inputs["cam_rot"][0], inputs["cam_trans"][0], inputs["intrinsics"][0]
        p2f = rasterize_fn(project_opencv(posed, cam_rot, cam_trans, K), image_size, image_size)
        n_posed = -face_normals(posed)                                                   # models/__init__.py:243
        out = {"output_normal": normal_image(p2f, n_posed @ cam_rot.t(), -1.0)}
        n_cano = face_normals(tri)                                                       # un-negated, :274
        for key, azim in (("normal_cano_front", 0.0), ("normal_cano_back", 180.0)):
            p2f = rasterize_fn(project_lookat(tri, azim, image_size), image_size, image_size, z_near=1.0)
            out[key] = normal_image(p2f, n_cano, 0.0)
        if not want_tri:
            return out, None
        if n_dev is not None:
            tri = tri[:min(int(n_dev.item()), F)]
    return out, tri
