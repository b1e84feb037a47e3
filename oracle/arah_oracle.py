"""ORACLE -- test infrastructure, NOT product code.

CPU (torch fp32) restatement of the reference's articulated-SDF volume-rendering hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module; the product (``arah_release_amd``) never does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function below against
fixtures in ``tests/golden/*.npz`` that were produced by importing the reference itself
(``tests/golden/make_golden.py``, run in the build container where /root/reference exists).
Third-party arithmetic that the reference delegates to pytorch3d 0.6.1 (``knn_points``, exact
1-NN, call sites ray_tracing.py:386,407) is restated as an exact 1-NN; its tie-breaking is
unpinned (no reference test covers it).

Every function cites the reference lines it follows; "RT" = im2mesh/metaavatar_render/renderer/
ray_tracing.py, "IDR" = .../implicit_differentiable_renderer.py, "RFU" = im2mesh/utils/
root_finding_utils.py (paths relative to the reference root).

All point sets are flat: (P,3).  The reference's batch dimension is folded into the ray
dimension (it flattens to (1,-1,3) itself, RT:178-184) with a per-ray camera origin.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch

KDTREE_WORKERS = -1      # threads of the exact 1-NN search (-1: all cores; bench.py's worker processes set their own share)
ROOT_THRESH = 1e-5       # RT:18, models/__init__.py:75
SPHERE_ITERS = 50        # RT:19
SURFACE_RANGE = 0.05     # RT:23
CLAMP_DIST = 0.1         # RT:174
BROYDEN_STEPS = 50       # broyden.py:4
BROYDEN_DVG = 1.0
BROYDEN_EPS = 1e-6


@dataclass
class Frame:
    """Everything that is constant for one temporal frame."""
    sdf_layers: List[Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]]
    skin_layers: List[Tuple[torch.Tensor, torch.Tensor]]
    color_layers: List[Tuple[torch.Tensor, torch.Tensor]]
    color_mode: str                    # 'no_view_dir' | 'idr'
    color_skips: Tuple[int, ...]
    multires_view: int
    pose_vec: Optional[torch.Tensor]   # (1, n_pose) constant tail of the colour input
    beta: float                        # |variance| before clipping
    verts: torch.Tensor                # (V,3) posed + trans
    vert_weights: torch.Tensor         # (V,24)
    bones: torch.Tensor                # (24,4,4)
    trans: torch.Tensor                # (3,)
    coord_min: float
    coord_max: float
    center: torch.Tensor               # (3,)
    counters: dict = field(default_factory=lambda: dict(n_sdf_fwd=0, n_sdf_grad=0, n_skin_fwd=0,
                                                        n_skin_jac=0, n_col=0, n_knn=0))

    @property
    def sdf_scale(self):
        # normalised SDF -> metres: sdf / 2 * 1.1 * (max - min)   (RT:528, IDR:359)
        return (self.coord_max - self.coord_min) * 1.1 / 2.0


# ----------------------------------------------------------------------------- coordinates
def normalize_points(fr, pts):
    """RFU:37-44."""
    rng = fr.coord_max - fr.coord_min
    p = pts - fr.center
    p = (p - fr.coord_min + rng * 0.05) / rng / 1.1
    return (p - 0.5) * 2.0


def unnormalize_points(fr, pts):
    """RFU:47-51."""
    rng = fr.coord_max - fr.coord_min
    return (pts / 2.0 + 0.5) * 1.1 * rng + fr.coord_min - rng * 0.05 + fr.center


# ----------------------------------------------------------------------------- networks
def sdf_features(fr, x):
    """FiLM-SIREN trunk h_k = sin(30 (f_k (W_k h + b_k) + phi_k)), k=1..6 -> (P,256)
    (hyperlayers.py:412-415, siren_modules.py:35-37)."""
    h = x
    for W, b, f, p in fr.sdf_layers[:-1]:
        h = torch.sin(30.0 * (f * (h @ W.t() + b) + p))
    return h


def sdf_forward(fr, x, count=True):
    """Normalised SDF value (P,) and feature (P,256) at normalised points x (P,3)."""
    if count:
        fr.counters["n_sdf_fwd"] += x.shape[0]
    h = sdf_features(fr, x)
    W, b, _, _ = fr.sdf_layers[-1]
    return (h @ W.t() + b)[:, 0], h


def sdf_forward_grad(fr, x):
    """SDF, feature and d sdf / d x by an explicit reverse sweep (what autograd does at IDR:336-338)."""
    fr.counters["n_sdf_fwd"] += x.shape[0]
    fr.counters["n_sdf_grad"] += x.shape[0]
    h = x
    dacts = []
    for W, b, f, p in fr.sdf_layers[:-1]:
        z = 30.0 * (f * (h @ W.t() + b) + p)
        dacts.append(torch.cos(z) * (30.0 * f))
        h = torch.sin(z)
    W, b, _, _ = fr.sdf_layers[-1]
    sdf = (h @ W.t() + b)[:, 0]
    g = W.expand(x.shape[0], -1)
    for (Wk, _, _, _), da in zip(reversed(fr.sdf_layers[:-1]), reversed(dacts)):
        g = (g * da) @ Wk
    return sdf, h, g


def skin_logits(fr, x):
    """Deformer MLP, Softplus(beta=100) (metaavatar/models/decoder.py:201-233)."""
    h = x
    n = len(fr.skin_layers)
    for i, (W, b) in enumerate(fr.skin_layers):
        h = h @ W.t() + b
        if i < n - 1:
            h = torch.nn.functional.softplus(h, beta=100)
    return h


def hierarchical_softmax(x):
    """25 logits -> 24 weights along the SMPL kinematic tree (utils/utils.py:138-181)."""
    sg = torch.sigmoid(x)
    sm = lambda idx: torch.softmax(x[:, idx], dim=-1)
    P = x.shape[0]
    w = [None] * 24
    root_split = sm([1, 2, 3])
    w[0] = 1.0 - sg[:, 0]
    for k, j in enumerate((1, 2, 3)):
        w[j] = sg[:, 0] * root_split[:, k]
    # a child takes sigmoid(gate) of its parent's mass, the parent keeps the rest
    chain = [((1, 2, 3), (4, 5, 6), (4, 5, 6)), ((4, 5, 6), (7, 8, 9), (7, 8, 9)),
             ((7, 8), (10, 11), (10, 11))]
    for parents, children, gates in chain:
        for p, c, gt in zip(parents, children, gates):
            w[c] = w[p] * sg[:, gt]
            w[p] = w[p] * (1.0 - sg[:, gt])
    spine_split = sm([12, 13, 14])
    for k, j in enumerate((12, 13, 14)):
        w[j] = w[9] * sg[:, 24] * spine_split[:, k]
    w[9] = w[9] * (1.0 - sg[:, 24])
    chain = [((12,), (15,), (15,)), ((13, 14), (16, 17), (16, 17)), ((16, 17), (18, 19), (18, 19)),
             ((18, 19), (20, 21), (20, 21)), ((20, 21), (22, 23), (22, 23))]
    for parents, children, gates in chain:
        for p, c, gt in zip(parents, children, gates):
            w[c] = w[p] * sg[:, gt]
            w[p] = w[p] * (1.0 - sg[:, gt])
    return torch.stack(w, dim=-1)


def query_weights(fr, x_hat, count=True):
    """Canonical (metric) points -> 24 skinning weights (RFU:54-113, 25-logit branch)."""
    if count:
        fr.counters["n_skin_fwd"] += x_hat.shape[0]
    return hierarchical_softmax(skin_logits(fr, normalize_points(fr, x_hat)) * 20.0)


def blend_transforms(fr, w):
    """(P,24) -> (P,4,4)  (RFU:26)."""
    return (w @ fr.bones.reshape(24, 16)).reshape(-1, 4, 4)


def lbs_forward(fr, x_hat, count=True):
    """x_bar = (sum_j w_j(x_hat) A_j) [x_hat;1]  (RFU:147-167, 13-34). Returns x_bar (P,3), T (P,4,4)."""
    T = blend_transforms(fr, query_weights(fr, x_hat, count))
    xb = torch.einsum("pij,pj->pi", T[:, :3, :3], x_hat) + T[:, :3, 3]
    return xb, T


def lbs_jacobian(fr, x_hat):
    """d x_bar / d x_hat (P,3,3) row by row with autograd (RFU:170-226, diff_operators.py:53-66)."""
    fr.counters["n_skin_jac"] += x_hat.shape[0]
    with torch.enable_grad():
        x = x_hat.detach().clone().requires_grad_(True)
        xb, _ = lbs_forward(fr, x, count=False)
        rows = [torch.autograd.grad(xb[:, i].sum(), x, retain_graph=(i < 2))[0] for i in range(3)]
    return torch.stack(rows, dim=1)


def positional_encoding(x, n_freqs):
    out = [x]
    for k in range(n_freqs):
        out += [torch.sin(x * float(2 ** k)), torch.cos(x * float(2 ** k))]
    return torch.cat(out, dim=-1)


def color_forward(fr, points, normals, view_dirs, feat):
    """RenderingNetwork (metaavatar_render/models/decoder.py:69-124)."""
    fr.counters["n_col"] += points.shape[0]
    if fr.multires_view > 0:
        view_dirs = positional_encoding(view_dirs, fr.multires_view)
    if fr.pose_vec is not None:
        feat = torch.cat([feat, fr.pose_vec.expand(feat.shape[0], -1)], dim=-1)
    if fr.color_mode == "idr":
        inp = torch.cat([points, view_dirs, normals, feat], dim=-1)
    elif fr.color_mode == "no_view_dir":
        inp = torch.cat([points, normals, feat], dim=-1)
    else:
        raise ValueError(fr.color_mode)
    x = inp
    n = len(fr.color_layers)
    for l, (W, b) in enumerate(fr.color_layers):
        if l in fr.color_skips:
            x = torch.cat([inp, x], dim=-1)
        x = x @ W.t() + b
        if l < n - 1:
            x = torch.relu(x)
    return torch.sigmoid(x)


# ----------------------------------------------------------------------------- nearest vertex
def nearest_vertex(fr, pts):
    """Exact 1-NN among the posed SMPL vertices (pytorch3d.ops.knn_points, RT:386,407)."""
    fr.counters["n_knn"] += pts.shape[0]
    if pts.shape[0] == 0:
        return torch.zeros(0, dtype=torch.long)
    if not hasattr(fr, "_kdtree"):
        from scipy.spatial import cKDTree
        fr._kdtree = cKDTree(fr.verts.double().numpy())
    _, idx = fr._kdtree.query(pts.double().numpy(), k=1, workers=KDTREE_WORKERS)
    return torch.from_numpy(np.ascontiguousarray(idx)).long()


def nn_inverse_lbs(fr, pts):
    """Inverse LBS with the nearest vertex's weights: returns raw canonical x_hat0 (P,3), T0 (P,4,4)
    (RT:382-400 without the final normalisation; RT:403-422)."""
    T = blend_transforms(fr, fr.vert_weights[nearest_vertex(fr, pts)])
    ph = torch.cat([pts - fr.trans, torch.ones(pts.shape[0], 1)], dim=-1)
    xh = torch.einsum("pij,pj->pi", torch.linalg.inv(T), ph)[:, :3]
    return xh, T


# ----------------------------------------------------------------------------- Broyden
def broyden(g, x0, T0, Jinv0, max_steps=BROYDEN_STEPS, cvg=ROOT_THRESH, dvg=BROYDEN_DVG, eps=BROYDEN_EPS):
    """Batched 'good Broyden' with per-point retirement and best-iterate tracking (broyden.py:4-78).

    g(x (n,D), ids (n,)) -> (gx (n,D), T (n,4,4)) evaluates the residual for the points ``ids``.
    Returns x_best (P,D), T_best (P,4,4), |g|_best (P,), converged (P,).
    """
    P, D = x0.shape
    x = x0.clone()
    T = T0.clone()
    Jinv = Jinv0.clone()
    everyone = torch.arange(P)
    gx, _ = g(x, everyone)                       # T of the initial evaluation is discarded (:35)
    x_best, T_best = x.clone(), T.clone()
    err_best = torch.linalg.norm(gx, dim=-1)
    act = everyone
    step = -torch.einsum("pij,pj->pi", Jinv, gx)
    for _ in range(max_steps):
        if act.numel() == 0:
            break
        dx = step
        x[act] = x[act] + dx
        g_new, T_new = g(x[act], act)
        dg = g_new - gx[act]
        gx[act] = gx[act] + dg
        T[act] = T_new
        err = torch.linalg.norm(gx[act], dim=-1)
        better = err < err_best[act]
        bi = act[better]
        err_best[bi] = err[better]
        x_best[bi] = x[bi]
        T_best[bi] = T[bi]
        keep = (err_best[act] > cvg) & (err < dvg)
        act, dx, dg = act[keep], dx[keep], dg[keep]
        if act.numel() == 0:
            break
        Ja = Jinv[act]
        vT = torch.einsum("pi,pij->pj", dx, Ja)
        a = dx - torch.einsum("pij,pj->pi", Ja, dg)
        b = (vT * dg).sum(-1, keepdim=True)
        b = torch.where(b >= 0, b + eps, b - eps)
        Ja = Ja + (a / b).unsqueeze(-1) * vT.unsqueeze(-2)
        Jinv[act] = Ja
        step = -torch.einsum("pij,pj->pi", Ja, gx[act])
    return x_best, T_best, err_best, err_best < cvg


# ----------------------------------------------------------------------------- loop A + B
def sphere_trace(fr, o, d, near, far):
    """50 nearest-vertex-skinned sphere-tracing steps (RT:174-241).
    Returns raw canonical x (N,3), T (N,4,4), depth t (N,), diverged (N,)."""
    assert bool((near <= far).all())
    N = o.shape[0]
    t = near.clone()
    unfinished = near < far
    diverged = near >= far
    x_cur = torch.zeros(N, 3)
    T_cur = torch.zeros(N, 4, 4)
    for _ in range(SPHERE_ITERS):
        idx = unfinished.nonzero()[:, 0]
        if idx.numel() == 0:
            break
        xh, T = nn_inverse_lbs(fr, o[idx] + t[idx, None] * d[idx])
        xn = normalize_points(fr, xh)
        sdf = sdf_forward(fr, xn)[0] * fr.sdf_scale
        x_cur[idx] = xn
        T_cur[idx] = T
        march = sdf.clamp(-CLAMP_DIST, CLAMP_DIST)
        upd = march.abs() > ROOT_THRESH
        t[idx] = torch.where(upd, t[idx] + march, t[idx])
        diverged[idx[upd]] = t[idx[upd]] >= far[idx[upd]]
        done = (sdf.abs() <= ROOT_THRESH) | diverged[idx]
        unfinished[idx[done]] = False
    return unnormalize_points(fr, x_cur), T_cur, t, diverged


def joint_root_find(fr, o, d, sel, x0, z0, T0):
    """4-D root find on u=(x_hat, depth) for the rays in ``sel`` (RFU:365-484).
    Returns x_hat (N,3) raw canonical, depth (N,), T (N,4,4), converged (N,)."""
    x_opt, z_opt, T_opt = x0.clone(), z0.clone(), T0.clone()
    conv = torch.zeros_like(sel)
    ids = sel.nonzero()[:, 0]
    if ids.numel() == 0:
        return x_opt, z_opt, T_opt, conv
    xs, ds, os_ = x0[ids], d[ids], o[ids]
    J = torch.zeros(ids.numel(), 4, 4)
    J[:, 1:, :3] = lbs_jacobian(fr, xs)                           # RFU:406
    # d(metric sdf)/d(metric x) == d(normalised sdf)/d(normalised x) up to rounding (RFU:408-413)
    with torch.enable_grad():
        xg = xs.detach().clone().requires_grad_(True)
        sdf = sdf_forward(fr, normalize_points(fr, xg))[0] * fr.sdf_scale
        fr.counters["n_sdf_grad"] += xs.shape[0]
        J[:, 0, :3] = torch.autograd.grad(sdf.sum(), xg)[0]
    J[:, 1:, 3] = -ds                                              # RFU:417
    Jinv = torch.linalg.inv(J)

    def resid(u, k):
        xh, z = u[:, :3], u[:, 3:]
        tgt = ds[k] * z + os_[k] - fr.trans
        xb, T = lbs_forward(fr, xh)
        sdf = sdf_forward(fr, normalize_points(fr, xh))[0] * fr.sdf_scale
        return torch.cat([sdf[:, None], xb - tgt], dim=-1), T

    u, T, _, ok = broyden(resid, torch.cat([xs, z0[ids, None]], dim=-1), T0[ids], Jinv)
    x_opt[ids], z_opt[ids], T_opt[ids], conv[ids] = u[:, :3], u[:, 3], T, ok
    return x_opt, z_opt, T_opt, conv


def trace_rays(fr, o, d, near, far, eval_mode=True):
    """sphere_tracing() of the reference incl. the post-processing (RT:174-296).
    Returns points_hat_norm (N,3), T (N,4,4), converged (N,), start (N,), end (N,)."""
    x_cur, T_cur, t, diverged = sphere_trace(fr, o, d, near, far)
    sel = ~diverged if eval_mode else torch.ones_like(diverged)
    x_opt, z_opt, T_opt, conv = joint_root_find(fr, o, d, sel, x_cur, t, T_cur)
    conv = conv & (z_opt >= near) & (z_opt <= far)                # RT:266
    start = torch.where(conv, z_opt, near)                         # RT:274-277
    return normalize_points(fr, x_opt), T_opt, conv, start, far.clone()


# ----------------------------------------------------------------------------- sampling + loop C
def perturb_z_vals(z, t_rand, fix_idx=None):
    """Stratified jitter inside the intervals between neighbouring samples (RT:298-311).  t_rand: the uniform draws the
    reference takes from torch.rand (same shape as z); fix_idx: a column that stays at its interval's midpoint (RT:305-307)."""
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mids, z[..., -1:]], dim=-1)
    lower = torch.cat([z[..., :1], mids], dim=-1)
    t = t_rand.clone()
    if fix_idx is not None:
        t[..., fix_idx] = 0.5
    return lower + (upper - lower) * t


def sample_depths(conv, start, end, near, n_steps, n_near, n_far, jitter=None):
    """Depth samples per ray (RT:313-350): returns z (N,S), mask (N,S).  ``jitter`` None: eval mode.  Training mode
    (RT:319-320, 332-333, 344-345): jitter = (rand_steps (N,S), rand_near (N,n_near+1), rand_far (N,n_far)), the three
    torch.rand draws of the reference in its order; the surface sample itself (column n_near // 2) is not perturbed."""
    N = start.shape[0]
    lin = torch.linspace(0.0, 1.0, n_steps, dtype=torch.float32)
    z = start[:, None] + (end - start)[:, None] * lin
    if jitter is not None:
        z = perturb_z_vals(z, jitter[0])
    mask = torch.ones(N, n_steps, dtype=torch.bool)
    if n_near > 0 or n_far > 0:
        lin_s = torch.linspace(0.0, 1.0, n_near + 1, dtype=torch.float32)
        zs = start[:, None] - SURFACE_RANGE + SURFACE_RANGE * 2 * lin_s
        if jitter is not None:
            zs = perturb_z_vals(zs, jitter[1], fix_idx=n_near // 2)
        n_c = n_near + 1
        block = zs
        if n_far > 0:
            lin_f = torch.linspace(0.0, 1.0, n_far, dtype=torch.float32)
            span = torch.maximum(start - SURFACE_RANGE - near, torch.tensor(1e-5))
            zf = near[:, None] + span[:, None] * lin_f
            if jitter is not None:
                zf = perturb_z_vals(zf, jitter[2])
            block = torch.sort(torch.cat([zs, zf], dim=-1), dim=-1)[0]
            n_c = n_near + 1 + n_far
        z[conv, :n_c] = block[conv]
        mask[conv, n_c:] = False
    return z, mask


def canonicalize(fr, pts):
    """Posed points (P,3) -> canonical by 3-D Broyden on forward LBS (RT:403-461, RFU:267-362).
    Returns x_hat_norm (P,3), T (P,4,4), converged (P,)."""
    if pts.shape[0] == 0:
        return torch.zeros(0, 3), torch.zeros(0, 4, 4), torch.zeros(0, dtype=torch.bool)
    x0, T0 = nn_inverse_lbs(fr, pts)
    tgt = pts - fr.trans
    # the weights at x0 give both J^-1_0 (RFU:327-328) and, through g(x0), the first residual
    Jinv = torch.linalg.inv(blend_transforms(fr, query_weights(fr, x0))[:, :3, :3])

    def resid(x, k):
        xb, T = lbs_forward(fr, x)
        return xb - tgt[k], T

    x, T, _, ok = broyden(resid, x0, T0, Jinv)
    return normalize_points(fr, x), T, ok


def sample_and_canonicalize(fr, o, d, conv, start, end, near, n_steps, n_near, n_far):
    """ray_sampler() (RT:313-380): returns pts (N,S,3), T (N,S,4,4), converged (N,S), z (N,S)."""
    z, mask = sample_depths(conv, start, end, near, n_steps, n_near, n_far)
    N, S = z.shape
    pts = o[:, None, :] + z[..., None] * d[:, None, :]
    xn, T, ok = canonicalize(fr, pts[mask])
    out_p = torch.zeros(N, S, 3)
    out_T = torch.zeros(N, S, 4, 4)
    out_m = torch.zeros(N, S, dtype=torch.bool)
    out_p[mask], out_T[mask], out_m[mask] = xn, T, ok
    return out_p, out_T, out_m, z


# ----------------------------------------------------------------------------- loop D
def shade_composite(fr, pts, z, T, mask, view_dirs, n_steps, cano_view_dirs, render_last_pt=False):
    """get_rbg_value_vol_sdf() in eval mode (IDR:261-396) for rays that own >= 1 valid sample.
    pts (n,S,3) normalised canonical, z (n,S), T (n,S,4,4), mask (n,S), view_dirs (n,3).
    Returns rgb (n,3), acc (n,1)."""
    n, S = z.shape
    lengths = mask.sum(-1)
    packed = torch.arange(S)[None, :] < lengths[:, None]          # left-packed slots (IDR:284-289)
    vp = pts[mask]
    vT = T[mask]
    vd = view_dirs[:, None, :].expand(n, S, 3)[mask]
    if cano_view_dirs:
        Rinv = torch.linalg.inv(vT)[:, :3, :3]
        vin = torch.einsum("pij,pj->pi", Rinv, -vd)
    else:
        vin = -vd
    sdf, feat, normal = sdf_forward_grad(fr, vp)
    if not cano_view_dirs:
        normal = torch.einsum("pij,pj->pi", vT[:, :3, :3], normal)   # IDR:340
    sdf = sdf * fr.sdf_scale
    rgb = color_forward(fr, vp, normal, vin, feat)
    beta = min(max(abs(fr.beta), 1e-6), 1e6)
    inv_beta = 1.0 / beta
    dens = torch.relu(inv_beta * (0.5 + 0.5 * torch.sign(-sdf) * (1 - torch.exp(-sdf.abs() * inv_beta))))
    rgb_s = torch.zeros(n, S, 3)
    den_s = torch.zeros(n, S)
    z_s = torch.full((n, S), 1e10)
    rgb_s[packed], den_s[packed], z_s[packed] = rgb, dens, z[mask]
    delta = z_s[:, 1:] - z_s[:, :-1]
    if render_last_pt:
        delta = torch.cat([delta, torch.full((n, 1), 1e10)], dim=-1)
    else:
        delta = torch.cat([delta, torch.full((n, 1), 1.0 / n_steps)], dim=-1)
        delta[torch.arange(n), lengths - 1] = 1.0 / n_steps
    alpha = 1.0 - torch.exp(-den_s * delta)
    trans = torch.cumprod(torch.cat([torch.ones(n, 1), 1.0 - alpha + 1e-7], dim=-1), dim=-1)[:, :-1]
    w = alpha * trans
    acc = (w * packed).sum(-1, keepdim=True).clamp(0, 1)
    return (rgb_s * (w * packed)[..., None]).sum(1), acc


# ----------------------------------------------------------------------------- whole renderer
def render(fr, o, d, near, far, n_steps=64, n_near=16, n_far=16, cano_view_dirs=True,
           render_last_pt=False, pose_R=None, pose_t=None, chunk=20480, return_intermediates=False):
    """IDHRNetwork.forward in eval mode (IDR:42-259) on flat rays. Returns dict of (N,..) tensors."""
    with torch.no_grad():
        xn, T_s, conv, start, end = trace_rays(fr, o, d, near, far)
        s_pts, s_T, s_mask, s_z = sample_and_canonicalize(fr, o, d, conv, start, end, near,
                                                          n_steps, n_near, n_far)
        vol = s_mask.any(-1)
        rgb = torch.zeros(o.shape[0], 3)
        acc = torch.zeros(o.shape[0])
        ids = vol.nonzero()[:, 0]
        for c in range(0, ids.numel(), chunk):
            k = ids[c:c + chunk]
            r, a = shade_composite(fr, s_pts[k], s_z[k], s_T[k], s_mask[k], d[k], n_steps,
                                   cano_view_dirs, render_last_pt)
            rgb[k], acc[k] = r, a[:, 0]
        pw = o + start[:, None] * d
        if pose_R is not None:
            pw = pw @ pose_R.t() + pose_t
        surface = conv & (xn.abs() <= 1.0).all(-1)
        pw = torch.where(surface[:, None], pw, torch.zeros_like(pw))
    out = {"points_cam": pw, "network_body_mask": vol, "rgb_values": rgb, "acc": acc}
    if return_intermediates:
        out.update(points_hat_norm=xn, surface_T=T_s, converged=conv, dists=start,
                   sampler_pts=s_pts, sampler_dists=s_z, sampler_transforms=s_T,
                   sampler_converge_mask=s_mask)
    return out


# ----------------------------------------------------------------------------- adapters
def frame_from_model(model, inputs):
    """Build a Frame from a (reference- or build-) MetaAvatarRender-like module and an input dict.
    Runs the per-frame hypernetwork with torch (it is outside the per-sample hot path)."""
    with torch.no_grad():
        dec_in = {"coords": torch.zeros(1, 1, 3), "rots": inputs["rots"][:1], "Jtrs": inputs["Jtrs"][:1]}
        if "geo_latent_code_idx" in inputs:
            dec_in["latent"] = model.latent(inputs["geo_latent_code_idx"])
        decoder = model.sdf_decoder(dec_in)["decoder"]
        sdf_layers = []
        for i in range(len(decoder) - 1):
            lin = decoder[i][0]
            sdf_layers.append((lin.weights[0].float(), lin.biases[0, 0].float(),
                               lin.freq.reshape(-1).float(), lin.phase_shift.reshape(-1).float()))
        sdf_layers.append((decoder[-1].weights[0].float(), decoder[-1].biases[0, 0].float(), None, None))

        def fold(lin):
            if hasattr(lin, "weight_g"):
                v = lin.weight_v
                return (lin.weight_g * v / v.norm(dim=1, keepdim=True)).detach().float(), lin.bias.detach().float()
            return lin.weight.detach().float(), lin.bias.detach().float()

        sk = model.skinning_model.skinning_decoder_fwd
        skin_layers = [fold(getattr(sk, "lin%d" % l)) for l in range(sk.num_layers - 1)]
        cd = model.color_decoder
        color_layers = [fold(getattr(cd, "lin%d" % l)) for l in range(cd.num_layers - 1)]
        pose_cond = dict(inputs["pose_cond"])
        if "latent_code_idx" in pose_cond:
            pose_cond["latent_code"] = model.latent(pose_cond["latent_code_idx"])
        t = cd.pose_encoder_type
        if t == "latent":
            pose_vec = pose_cond["latent_code"]
        elif t == "leap":
            pose_vec = cd.pose_encoder(pose_cond["rots_full"][:1], pose_cond["Jtrs_posed"][:1])
        elif t in ("root", "hybrid"):
            pose_vec = torch.cat([pose_cond["rots_full"][:1, :1].reshape(1, 9),
                                  pose_cond["Jtrs_posed"][:1, :1].reshape(1, 3)], dim=-1)
            if t == "hybrid":
                pose_vec = torch.cat([pose_vec, pose_cond["latent_code"]], dim=-1)
        else:
            pose_vec = None
        embedview = getattr(cd, "embedview_fn", None)
        multires_view = getattr(cd, "multires_view", None)
        if multires_view is None:
            multires_view = 0 if embedview is None else (cd.lin0.weight_v.shape[1] - 6 - 256 -
                                                         (0 if pose_vec is None else pose_vec.shape[1]) - 3) // 6
        return Frame(sdf_layers=sdf_layers, skin_layers=skin_layers, color_layers=color_layers,
                     color_mode=cd.mode, color_skips=tuple(cd.skips), multires_view=int(multires_view),
                     pose_vec=None if pose_vec is None else pose_vec.float(),
                     beta=float(torch.linalg.norm(model.deviation_decoder.variance)),
                     verts=inputs["smpl_verts"][0].float(), vert_weights=inputs["skinning_weights"][0].float(),
                     bones=inputs["bone_transforms"][0].float(), trans=inputs["trans"][0, 0].float(),
                     coord_min=float(inputs["coord_min"].reshape(-1)[0]),
                     coord_max=float(inputs["coord_max"].reshape(-1)[0]),
                     center=inputs["center"][0, 0].float())


def render_inputs(model, inputs, cano_view_dirs, n_steps=64, n_near=16, n_far=16, render_last_pt=False,
                  return_intermediates=False):
    """Oracle counterpart of MetaAvatarRender.forward(inputs, eval=True) for B == 1."""
    fr = frame_from_model(model, inputs)
    B, N, _ = inputs["ray_dirs"].shape
    o = inputs["cam_loc"].reshape(B, 1, 3).expand(B, N, 3).reshape(-1, 3).float()
    d = inputs["ray_dirs"].reshape(-1, 3).float()
    nf = inputs["body_bounds_intersections"].reshape(-1, 2).float()
    pose = inputs["pose"][0].float()
    out = render(fr, o, d, nf[:, 0].contiguous(), nf[:, 1].contiguous(), n_steps, n_near, n_far,
                 cano_view_dirs, render_last_pt, pose[:3, :3], pose[:3, 3],
                 return_intermediates=return_intermediates)
    out["frame"] = fr
    return out
