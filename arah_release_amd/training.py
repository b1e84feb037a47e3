"""Differentiable (PyTorch autograd, on the GPU) half of the training step.

In the reference, loops A-C of the renderer run under ``torch.no_grad()`` even while training
(implicit_differentiable_renderer.py:87, root_finding_utils.py:302,348,460): they are served by the
HIP kernels unchanged.  What carries gradients is

  * loop D on the 2048 sampled rays: SDF value / feature, its input gradient (double backward through the
    SIREN), the colour MLP, the VolSDF density and the compositing (IDR:261-396, training branches),
    including the implicit-gradient re-attachment of the canonical points to the skinning network
    (IDR:315-334);
  * the small regulariser queries (skinning weights at surface points, SDF inside the body, eikonal and
    uniform off-surface points, IDR:73-79,117-140);
  * the loss (renderer/loss.py:123-191).

These are kept on autograd, as SURVEY 7 step 7 plans for a first cut; a hand-written backward of
loop D is the follow-up.  Everything here is plain torch on whatever device the tensors live on.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .tall import gram, gram_grouped, mv3


# ------------------------------------------------------------------------------------------------
# canonical-space helpers (root_finding_utils.py:13-113)
# ------------------------------------------------------------------------------------------------
def normalize_canonical_points(pts, coord_min, coord_max, center):
    span = coord_max - coord_min
    return (((pts - center) - coord_min + span * 0.05) / span / 1.1 - 0.5) * 2.0


def unnormalize_canonical_points(pts, coord_min, coord_max, center):
    span = coord_max - coord_min
    return (pts / 2.0 + 0.5) * 1.1 * span + coord_min - span * 0.05 + center


def _hsoftmax_paths():
    """The factors each of the 24 weights is a product of, found by running the reference's recursion
    (utils/utils.py:138-181) on index lists: columns of F = [gate(25) | 1 - gate(25) | softmax(x[1:4]) | softmax(x[12:15]) | 1]."""
    G, K, HIPS, CHEST, ONE = 0, 25, 50, 53, 56
    w = [None] * 24
    w[0] = [K + 0]
    for k in range(3):
        w[1 + k] = [G + 0, HIPS + k]

    def hand_down(parent, child, g):
        w[child] = w[parent] + [G + g]
        w[parent] = w[parent] + [K + g]

    for parent, child in ((1, 4), (2, 5), (3, 6), (4, 7), (5, 8), (6, 9), (7, 10), (8, 11)):
        hand_down(parent, child, child)
    for k in range(3):
        w[12 + k] = w[9] + [G + 24, CHEST + k]
    w[9] = w[9] + [K + 24]
    for parent, child in ((12, 15), (13, 16), (14, 17), (16, 18), (17, 19), (18, 20), (19, 21), (20, 22), (21, 23)):
        hand_down(parent, child, child)
    depth = max(len(p) for p in w)
    return [p + [ONE] * (depth - len(p)) for p in w]


_HSOFTMAX_PATHS = {}


def hierarchical_softmax(x):
    """(..., 25) logits -> (..., 24) weights along the SMPL kinematic tree (utils/utils.py:138-181).
    The reference hands weight down the tree joint by joint (some eighty element-wise launches, two hundred in backward);
    every weight is a fixed product of gates / complements / softmax entries, so here it is one gather and one product."""
    lead = x.shape[:-1]
    x = x.reshape(-1, 25)
    gate = torch.sigmoid(x)
    F_ = torch.cat([gate, 1.0 - gate, torch.softmax(x[:, 1:4], dim=-1), torch.softmax(x[:, 12:15], dim=-1),
                    torch.ones_like(x[:, :1])], dim=-1)
    key = (str(x.device), x.dtype)
    if key not in _HSOFTMAX_PATHS:
        paths = torch.tensor(_hsoftmax_paths(), dtype=torch.long)              # (24, depth)
        sel = torch.zeros(F_.shape[1], paths.numel(), dtype=x.dtype)
        sel[paths.t().reshape(-1), torch.arange(paths.numel())] = 1.0           # column c of the product = column idx[c] of F
        _HSOFTMAX_PATHS[key] = (sel.to(x.device), paths.shape[1])
    sel, depth = _HSOFTMAX_PATHS[key]
    # the gather as a product with a one-hot matrix: exact (every sum has one non-zero term), and its backward is a GEMM --
    # index_select's is an atomic index_add (1.4 ms per step on the 1.3e5-row call), advanced indexing's a sort (0.9 ms)
    cols = torch.matmul(F_, sel).reshape(-1, depth, 24).unbind(1)
    out = cols[0]
    for c in cols[1:]:                                             # left to right, the order the recursion multiplies in
        out = out * c
    return out.reshape(*lead, 24)


class _HSoftmaxOp(torch.autograd.Function):
    """hierarchical_softmax(scale * logits) as one launch each way (hip.hsoftmax_train_forward / _backward, csrc: k_hsoftmax_train)
    for the skinning queries of a training step on the device; first-order autograd only."""

    @staticmethod
    def forward(ctx, logits, scale):
        from . import hip
        x = logits.reshape(-1, 25).contiguous()
        ctx.save_for_backward(x)
        ctx.scale = scale
        return hip.hsoftmax_train_forward(x, scale).reshape(*logits.shape[:-1], 24)

    @staticmethod
    def backward(ctx, g):
        from . import hip
        (x,) = ctx.saved_tensors
        return hip.hsoftmax_train_backward(x, ctx.scale, g.reshape(-1, 24).contiguous()).reshape(*g.shape[:-1], 25), None


def query_weights(x_hat, coord_min, coord_max, center, skinning_model):
    """(B,N,3) canonical points -> (B,N,24) skinning weights (root_finding_utils.py:54-113, 25-logit branch)."""
    logits = skinning_model.decode_w(normalize_canonical_points(x_hat, coord_min, coord_max, center),
                                     c=torch.empty(x_hat.shape[0], 0, device=x_hat.device))
    if logits.shape[-1] != 25:
        raise ValueError("Wrong output size of skinning network. Expected 25, got %d." % logits.shape[-1])
    if logits.is_cuda and logits.dtype == torch.float32 and os.environ.get("ARAH_TRAIN_HSOFTMAX_OP", "1") != "0":
        return _HSoftmaxOp.apply(logits, 20.0)
    return hierarchical_softmax(logits * 20.0)


def forward_skinning(x_hat, coord_min, coord_max, center, skinning_model, bone_transforms):
    """x_bar = (sum_j w_j A_j) [x_hat; 1] (root_finding_utils.py:147-167, 13-34)."""
    w = query_weights(x_hat, coord_min, coord_max, center, skinning_model)
    T = torch.matmul(w, bone_transforms.reshape(bone_transforms.shape[0], 24, 16)).reshape(*w.shape[:2], 4, 4)
    x_bar = mv3(T[..., :3, :3], x_hat) + T[..., :3, 3]
    return x_bar, T


def input_jacobian(y, x):
    """d y / d x for y (1,P,3) computed from x (1,P,3), row by row, detached (diff_operators.py:53-66)."""
    rows = [torch.autograd.grad(y[..., i].sum(), x, retain_graph=True)[0] for i in range(y.shape[-1])]
    return torch.stack(rows, dim=-2)


# ------------------------------------------------------------------------------------------------
# loop D with gradients (IDR:261-396, self.training branches)
# ------------------------------------------------------------------------------------------------
def colsum(a):
    """a.sum(0) for a (P, n) stream of the training step: one pass at HBM speed on the device (hip.colsum), torch elsewhere."""
    if a.is_cuda and a.dtype == torch.float32:
        from . import hip
        return hip.colsum(a)
    return a.sum(0)


def _siren_grads(st, g_sdf, feat, sdf_w, sdf_b, freq, phase):
    """Gradients of the emitted SIREN's 7 weights, 7 biases, FiLM frequencies and phases from the operand streams of the training
    kernel (hip.shade_train_backward / sdf_normal_backward).  dW_k = adj(v_k)^T h_{k-1} + adj(vd_k)^T hd_{k-1}: the two pairs of a
    layer lie next to each other (hip._sdf_streams), so a layer is ONE product over 2 n rows and layers 2..6 ONE batched split-K
    launch.  db_k = sum adj v_k needs no pass over the samples at all: adj v_k = 30 f_k zb_k and the kernel already reduces
    d phi_k = sum 30 zb_k (train.hpp, step 3), so db_k = f_k * d phi_k."""
    avv, hh, h0 = st["avv"], st["hh"], st["h0"]
    G, _, R, W = avv.shape
    grads = []
    n_in = sdf_w[0].shape[-1]
    grads.append(gram(h0.reshape(2 * R, 4)[:, :n_in], avv[0].reshape(2 * R, W)).t().reshape(sdf_w[0].shape))
    hidden = gram_grouped(avv[1:].reshape(G - 1, 2 * R, W), hh.reshape(G - 1, 2 * R, W))
    grads += [hidden[k].reshape(sdf_w[k + 1].shape) for k in range(G - 1)]
    grads.append((gram(g_sdf.reshape(-1, 1), feat) + colsum(st["hd"][6]).unsqueeze(0)).reshape(sdf_w[6].shape))
    db = st["film_phase"] * freq.reshape(st["film_phase"].shape)
    grads += [db[k].reshape(sdf_b[k].shape) for k in range(6)]
    grads.append(g_sdf.sum().reshape(sdf_b[6].shape))
    grads.append(st["film_freq"].reshape(freq.shape))
    grads.append(st["film_phase"].reshape(phase.shape))
    return grads


class ShadeSamples(torch.autograd.Function):
    """Per-sample part of loop D as ONE custom op on the HIP kernels (csrc/train.hpp): SDF value, normal, colour in
    the forward; in the backward the kernel recomputes the forward, sweeps the colour MLP and the SIREN backwards
    (second-order path through the normal included) and streams the operands of the weight-gradient outer products,
    which are finished here with library GEMMs.  Replaces `sdf_network[:-1](x)`, `gradient(sdf, x)` with
    create_graph=True and `rendering_network(...)` of the reference (IDR:336-361) and their autograd graphs.

    apply(meta, x, *params) with meta = dict(frame, ws, T, view, view_orig, rotate_normal, ray_augm, mode, n_pose) and
    params = 7 SDF weights (out,in), 7 SDF biases, freq (6*256), phase (6*256), 6 folded colour weights, 6 colour
    biases, pose vector (n_pose,) -- passed so that autograd knows them; their VALUES are the ones packed in `frame`.
    Returns sdf (P,) in normalised units and rgb (P,3)."""

    @staticmethod
    def forward(ctx, meta, x, *params):
        from . import hip
        # keep: the forward call leaves the colour MLP's activations for the backward (hip.shade_train_forward)
        keep = os.environ.get("ARAH_TRAIN_HANDOVER", "1") != "0"
        res = hip.shade_train_forward(meta["frame"], meta["ws"], x, meta["T"], meta["view"], meta["view_orig"],
                                      meta["rotate_normal"], meta["ray_augm"], keep=keep)
        sdf, rgb = res[0], res[1]
        ctx.kept = res[2] if keep else None
        ctx.meta = meta
        ctx.save_for_backward(x, *params)
        return sdf, rgb

    @staticmethod
    def backward(ctx, g_sdf, g_rgb):
        from . import hip
        meta = ctx.meta
        x, *params = ctx.saved_tensors
        st = hip.shade_train_backward(meta["frame"], meta["ws"], x, meta["T"], meta["view"], meta["view_orig"],
                                      meta["rotate_normal"], meta["ray_augm"], g_sdf.contiguous(), g_rgb.contiguous(),
                                      kept=ctx.kept)
        sdf_w, sdf_b = params[0:7], params[7:14]
        col_w, col_b, pose = params[16:22], params[22:28], params[28]
        feat = st["cin"][:, :256]
        grads = _siren_grads(st, g_sdf, feat, sdf_w, sdf_b, params[14], params[15])
        # ---- colour MLP: dW_l = delta_l^T X_l; stream columns are [feat | x | n | PE(view) | 0]
        idr = meta["mode"] == "idr"
        n_pose = meta["n_pose"]
        cin, c, d = st["cin"], st["c"], st["d"]
        n_view = 27 if idr else 0

        def to_reference_columns(m, tail=None):   # (out, kin) in stream order -> (out, in_dim [+128]) in the reference's
            parts = [m[:, 256:259]]
            if idr:
                parts.append(m[:, 262:289])
            parts += [m[:, 259:262], m[:, :256]]
            return parts

        # the four 256-wide deltas travel as one buffer (hip.shade_train_backward: dd = delta_1, delta_4, delta_0, delta_3; cc = c_1,
        # c_4, c_2, c_5): their column sums are one reduction, delta_1^T c_1 and delta_4^T c_4 one batched product
        s1, s4, s0, s3 = (colsum(st["dd"][i]) for i in range(4))   # padding rows are zero
        g14 = gram_grouped(st["dd"][0:2], st["cc"][0:2])
        # (whole padded buffers: their row count is a multiple of the split-K chunking, no remainder product)
        dd, cc, cin_p = st["dd"], st["cc"], st["cin_pad"]
        m0 = to_reference_columns(gram(dd[2], cin_p))
        m3 = to_reference_columns(gram(dd[3], cin_p))
        if n_pose:
            m0.append(torch.outer(s0, pose.reshape(-1)))
            m3.append(torch.outer(s3, pose.reshape(-1)))
        m3.append(gram(dd[3], st["c2_pad"]))
        gw = [torch.cat(m0, dim=1), g14[0], gram(st["d2_pad"], cc[2]), torch.cat(m3, dim=1), g14[1],
              gram(d[5][:, :3], c[4])]
        gb = [s0, s1, colsum(st["d2_pad"]), s3, s4, colsum(d[5][:, :3])]
        grads += [g.reshape(w.shape) for g, w in zip(gw, col_w)]
        grads += [g.reshape(b.shape) for g, b in zip(gb, col_b)]
        if n_pose:
            p0 = 3 + n_view + 3 + 256
            grads.append((col_w[0][:, p0:p0 + n_pose].t() @ s0 + col_w[3][:, p0:p0 + n_pose].t() @ s3).reshape(pose.shape))
        else:
            grads.append(None)
        return (None, st["gx4"][:, :3]) + tuple(grads)


class SdfNormal(torch.autograd.Function):
    """SDF value and gradient at query points as one custom op on the training kernel without its colour half: what
    `sdf_network(x)` + `autograd.grad(sdf, x, create_graph=True)` give the regularisers (IDR:104-128: eikonal on the gradient,
    off-surface / inside terms on the value), with the second-order path into every SDF parameter in the backward.
    apply(meta, x, *params): meta = dict(frame, ws), params = 7 weights, 7 biases, freq, phase (as in ShadeSamples)."""

    @staticmethod
    def forward(ctx, meta, x, *params):
        from . import hip
        sdf, normal = hip.sdf_normal_forward(meta["frame"], meta["ws"], x)
        ctx.meta = meta
        ctx.save_for_backward(x, *params)
        return sdf, normal

    @staticmethod
    def backward(ctx, g_sdf, g_n):
        from . import hip
        x, *params = ctx.saved_tensors
        g_sdf, g_n = g_sdf.contiguous(), g_n.contiguous()
        st = hip.sdf_normal_backward(ctx.meta["frame"], ctx.meta["ws"], x, g_sdf, g_n)
        sdf_w, sdf_b = params[0:7], params[7:14]
        grads = _siren_grads(st, g_sdf, st["feat"], sdf_w, sdf_b, params[14], params[15])
        return (None, st["gx4"][:, :3]) + tuple(grads)


def _first(w):
    """w[0] of an emitted (1, out, in) weight as a VIEW whose backward is a view too (a select's backward zero-fills a new tensor
    and copies into it: three launches per emitted tensor and op, 37 per step)."""
    return w.reshape(w.shape[1:]) if w.shape[0] == 1 else w[0]


def sdf_normal_hip(frame, ws, sdf_network, x):
    """x (P,3) normalised -> sdf (P,1) in normalised units, d sdf / d x (P,3); differentiable w.r.t. the emitted network."""
    n = len(sdf_network)
    sdf_w = [_first(sdf_network[i][0].weights) for i in range(n - 1)] + [_first(sdf_network[n - 1].weights)]
    sdf_b = [sdf_network[i][0].biases.reshape(-1) for i in range(n - 1)] + [sdf_network[n - 1].biases.reshape(-1)]
    freq = torch.cat([sdf_network[i][0].freq.reshape(-1) for i in range(n - 1)])
    phase = torch.cat([sdf_network[i][0].phase_shift.reshape(-1) for i in range(n - 1)])
    sdf, normal = SdfNormal.apply(dict(frame=frame, ws=ws), x.contiguous(), *(sdf_w + sdf_b + [freq, phase]))
    return sdf.unsqueeze(-1), normal


def shade_samples_hip(idhr, frame, ws, sdf_network, x, T, view, view_orig, pose_cond, ray_augm):
    """x (P,3) -> sdf (P,1) normalised units, rgb (P,3), differentiable w.r.t. x and every network parameter."""
    from .nets import folded_weight
    rn = idhr.rendering_network
    n = len(sdf_network)
    sdf_w = [_first(sdf_network[i][0].weights) for i in range(n - 1)] + [_first(sdf_network[n - 1].weights)]
    sdf_b = [sdf_network[i][0].biases.reshape(-1) for i in range(n - 1)] + [sdf_network[n - 1].biases.reshape(-1)]
    freq = torch.cat([sdf_network[i][0].freq.reshape(-1) for i in range(n - 1)])
    phase = torch.cat([sdf_network[i][0].phase_shift.reshape(-1) for i in range(n - 1)])
    col_w = [folded_weight(getattr(rn, "lin%d" % l)) for l in range(rn.num_layers - 1)]
    col_b = [getattr(rn, "lin%d" % l).bias for l in range(rn.num_layers - 1)]
    pose = rn.pose_vector(pose_cond)
    n_pose = 0 if pose is None else int(pose.numel())
    meta = dict(frame=frame, ws=ws, T=T.detach().contiguous() if T is not None else None, view=view.detach().contiguous(),
                view_orig=view_orig.detach().contiguous() if view_orig is not None else None,
                rotate_normal=not idhr.cano_view_dirs, ray_augm=bool(ray_augm), mode=rn.mode, n_pose=n_pose)
    params = sdf_w + sdf_b + [freq, phase] + col_w + col_b + [pose.reshape(-1) if n_pose else x.new_zeros(1)]
    sdf, rgb = ShadeSamples.apply(meta, x.contiguous(), *params)
    return sdf.unsqueeze(-1), rgb


class CompositeSamples(torch.autograd.Function):
    """VolSDF density and alpha compositing over the compacted samples of every ray as one launch each way
    (arah_composite_train_forward / _backward, csrc/train.hpp) instead of ~40 element-wise launches and ~100 in backward.
    apply(lengths int32 (R,), offsets int64 (R,), z (P,), n_steps, render_last_pt, sdf (P,), rgb (P,3), inv_beta (1,))
    -> rgb_map (R,3), acc (R,)."""

    @staticmethod
    def forward(ctx, lengths, offsets, z, n_steps, render_last_pt, sdf, rgb, inv_beta):
        from . import hip
        sdf, rgb, inv_beta = sdf.contiguous(), rgb.contiguous(), inv_beta.reshape(1).contiguous()
        out = hip.composite_train_forward(lengths, offsets, sdf, rgb, z, inv_beta, n_steps, render_last_pt)
        ctx.save_for_backward(lengths, offsets, z, sdf, rgb, inv_beta)
        ctx.cfg = (n_steps, render_last_pt)
        return out

    @staticmethod
    def backward(ctx, g_map, g_acc):
        from . import hip
        lengths, offsets, z, sdf, rgb, inv_beta = ctx.saved_tensors
        g_sdf, g_rgb, g_ib = hip.composite_train_backward(lengths, offsets, sdf, rgb, z, inv_beta, ctx.cfg[0], ctx.cfg[1],
                                                          g_map.contiguous(), g_acc.contiguous())
        return None, None, None, None, None, g_sdf, g_rgb, g_ib


def shade_composite_train(idhr, sdf_network, points, z_vals, transforms_fwd, converge_mask, view_dirs,
                          view_dirs_orig, pose_cond, bone_transforms, coord_min, coord_max, center, n_steps,
                          ray_augm=False, point_batch_size=100000, frame=None, ws=None):
    """frame / ws given (GPU training): the per-sample networks run in the hand-written HIP forward / backward
    (ShadeSamples); otherwise plain autograd (the restatement the HIP op is tested against)."""
    n_rays, S, _ = points.shape
    dev = points.device
    lengths = converge_mask.sum(-1)
    slot = torch.arange(S, device=dev)[None, :] < lengths[:, None]          # left-packed valid slots
    # ONE compaction (a device -> host round trip for the count) serves every per-sample gather below and the scatter back
    # into left-packed [ray, slot] form; the reference indexes with the boolean mask each time (same elements, same order)
    ridx, sidx = converge_mask.nonzero(as_tuple=True)
    flat = ridx * S + (converge_mask.cumsum(-1) - 1)[ridx, sidx]             # k-th valid sample of a ray -> slot k
    pts = points[ridx, sidx]
    n_valid = pts.shape[0]
    far = None
    from .nets import SingleVarianceNetwork
    if (frame is not None and n_valid > 0 and isinstance(idhr.deviation_network, SingleVarianceNetwork)
            and os.environ.get("ARAH_TRAIN_LAZY", "1") != "0"):
        # Round 6: the per-sample networks run only where a gradient can flow.  A valid sample with metric sdf s > 0 has density
        # ib * (1/2 - 1/2 (1 - e)), e = exp(-s ib) (IDR:366-368).  Once s ib > 103.98 that exponential is EXACTLY 0 in fp32 (below
        # the smallest denormal), and so is every term it feeds: the density and its derivatives with respect to s and to beta
        # (both carry the factor e), hence alpha, the sample's weight, the gradient of its colour (weight x upstream) and of its
        # SDF value -- the sample's whole contribution to the forward AND to every parameter gradient is exactly zero, whatever
        # its normal and colour are.  (The eval forward's lazy shading stops at 17.3 beta, where the VALUE rounds to zero; the
        # derivative needs the exponential itself to vanish.)  One SDF forward over the valid samples (0.2 ms) finds them: 76-87 %
        # of a training view's samples at beta = 1e-3 (`tools/probes/train_far_samples.py`); the hand-written op, the
        # re-attachment to the skinning network and their weight-gradient products then see the other 13-24 %.  The far samples
        # still take part in the compositing with their SDF value (delta chains, the 1e-7 factors), exactly as before.
        # kFarCut = 110 leaves a margin for the difference between this pass's SDF value and the shading kernel's own (the same
        # f16-split trunk).  A learned beta that grows widens the band by itself; ARAH_TRAIN_LAZY=0 shades every valid sample.
        from . import hip
        with torch.no_grad():
            s_pre = hip.sdf_eval(frame, ws, pts)[0]
            to_m = (coord_max.reshape(-1)[0] - coord_min.reshape(-1)[0]) * (1.1 / 2.0)
            ib_pre = torch.reciprocal(torch.linalg.norm(idhr.deviation_network.variance).clip(1e-6, 1e6))
            kidx = (~(s_pre * to_m * ib_pre > 110.0)).nonzero().squeeze(1)          # (a NaN value stays)
        if kidx.numel() < n_valid:
            far = (kidx, s_pre)
            pts = pts[kidx]
    sub = (lambda t: t[far[0]]) if far is not None else (lambda t: t)
    Tf = sub(transforms_fwd[ridx, sidx])
    vd = sub(view_dirs[ridx])
    vd0 = sub(view_dirs_orig[ridx])
    if idhr.cano_view_dirs:
        Rb = torch.linalg.inv(Tf).detach()[:, :3, :3]
        vin = mv3(Rb, -vd)
        vin0 = mv3(Rb, -vd0)
    else:
        vin, vin0 = -vd, -vd0
    sdf_all, rgb_all = [], []
    if frame is not None:
        point_batch_size = max(point_batch_size, pts.shape[0])   # the HIP op streams its operands: no chunking needed
    for c in range(0, pts.shape[0], point_batch_size):
        pi = pts[c:c + point_batch_size].unsqueeze(0).requires_grad_(True)
        vi, vi0, Ti = vin[c:c + point_batch_size], vin0[c:c + point_batch_size], Tf[c:c + point_batch_size]
        with torch.enable_grad():
            if idhr.train_skinning_net and frame is not None:
                # Same re-attachment (IDR:315-334) with the Jacobian from the HIP seam (forward-mode tangents through the
                # skinning MLP, arah_skin_jacobian) instead of three autograd passes: d x_hat = -J^-1 d LBS only needs
                # the graph from the skinning parameters to x_lbs, the terms in d(pi) cancel identically.
                from . import hip
                pd = pi.detach()
                x_hat = unnormalize_canonical_points(pd, coord_min, coord_max, center)
                x_lbs, _ = forward_skinning(x_hat, coord_min, coord_max, center, idhr.skinning_model, bone_transforms)
                span = (coord_max.reshape(-1)[0] - coord_min.reshape(-1)[0]) * 1.1 / 2.0     # d x_hat / d pi
                J = hip.skin_jacobian(frame, ws, x_hat[0])
                if os.environ.get("ARAH_TRAIN_INV3", "1") != "0":
                    # (J span)^-1 = J^-1 / span: cofactors in one launch (the batched LU of torch.linalg.inv is four kernels and
                    # 0.5 ms for these 1.2e5 well-conditioned matrices); span stays on the device
                    Jinv = (hip.inverse3x3(J) / span).unsqueeze(0)
                else:
                    Jinv = torch.linalg.inv(J * span).unsqueeze(0)
                pi = pd - mv3(Jinv, x_lbs - x_lbs.detach())
            elif idhr.train_skinning_net:
                # x_hat is a root of LBS(x_hat) = x_bar found without a graph; re-attach it with the implicit
                # function theorem: d x_hat = -J^-1 d LBS  (IDR:315-334)
                x_hat = unnormalize_canonical_points(pi, coord_min, coord_max, center)
                x_lbs, _ = forward_skinning(x_hat, coord_min, coord_max, center, idhr.skinning_model, bone_transforms)
                Jinv = torch.linalg.inv(input_jacobian(x_lbs, pi)).detach()
                pi = pi - mv3(Jinv, x_lbs - x_lbs.detach())
            if frame is not None:
                sdf, rgb = shade_samples_hip(idhr, frame, ws, sdf_network, pi.squeeze(0), Ti, vi, vi0, pose_cond, ray_augm)
                sdf_all.append(sdf / 2.0 * 1.1 * (coord_max.squeeze() - coord_min.squeeze()))
                rgb_all.append(rgb)
                continue
            feat = sdf_network[:-1](pi).squeeze(0)
            sdf = sdf_network[-1](feat)
            normal = torch.autograd.grad(sdf, pi, torch.ones_like(sdf), create_graph=True, retain_graph=True)[0]
            if not idhr.cano_view_dirs:
                normal = mv3(Ti[:, :3, :3].unsqueeze(0), normal)
            if ray_augm:
                with torch.no_grad():   # keep the un-rotated view where the augmented one looks at the back face
                    nn_ = normal / torch.linalg.norm(normal, dim=-1, keepdim=True)
                    back = torch.arccos((nn_.squeeze(0) * vi).sum(-1)) >= np.pi / 2.0
                vi = torch.where(back[:, None], vi0, vi)
        sdf_all.append((sdf / 2.0 * 1.1 * (coord_max.squeeze() - coord_min.squeeze())).squeeze(0))
        rgb_all.append(idhr.rendering_network(pi.squeeze(0), normal.squeeze(0), vi, feat, pose_cond))
    sdf_v = torch.cat(sdf_all, dim=0) if sdf_all else pts.new_zeros(0, 1)
    rgb_v = torch.cat(rgb_all, dim=0) if rgb_all else pts.new_zeros(0, 3)
    if far is not None:   # back among all valid samples: the far ones with the pre-pass's SDF value (no graph) and a zero colour
        kidx, s_pre = far
        sdf_far = (s_pre / 2.0 * 1.1 * (coord_max.squeeze() - coord_min.squeeze())).reshape(-1)
        sdf_v = sdf_far.index_copy(0, kidx, sdf_v.reshape(-1)).unsqueeze(-1)
        rgb_v = torch.zeros(n_valid, 3, device=dev, dtype=rgb_v.dtype).index_copy(0, kidx, rgb_v)
    if (frame is not None and isinstance(idhr.deviation_network, SingleVarianceNetwork)
            and os.environ.get("ARAH_TRAIN_COMPOSITE_OP", "1") != "0"):
        # density + compositing as one op on the compacted samples (rays own contiguous runs of them: nonzero() is row-major);
        # beta is one number (decoder.py:127-133: ones_like(x) * |variance|)
        inv_beta = torch.reciprocal(torch.linalg.norm(idhr.deviation_network.variance).clip(1e-6, 1e6))
        len32 = lengths.to(torch.int32)
        offsets = torch.cumsum(lengths, 0) - lengths
        rgb_map, acc = CompositeSamples.apply(len32, offsets, z_vals[ridx, sidx].contiguous(), n_steps, idhr.render_last_pt,
                                              sdf_v.reshape(-1), rgb_v, inv_beta.reshape(1))
        return rgb_map, acc.unsqueeze(-1)
    beta = idhr.deviation_network(sdf_v).clip(1e-6, 1e6)
    inv_beta = torch.reciprocal(beta)
    dens_v = F.relu(inv_beta * (0.5 + 0.5 * torch.sign(-sdf_v) * (1 - torch.exp(-sdf_v.abs() * inv_beta))))
    rgb = torch.zeros(n_rays * S, 3, device=dev).index_copy(0, flat, rgb_v).reshape(n_rays, S, 3)
    dens = torch.zeros(n_rays * S, device=dev).index_copy(0, flat, dens_v.squeeze(-1)).reshape(n_rays, S)
    zp = torch.full((n_rays * S,), 1e10, device=dev).index_copy(0, flat, z_vals[ridx, sidx]).reshape(n_rays, S)
    delta = zp[:, 1:] - zp[:, :-1]
    if idhr.render_last_pt:
        delta = torch.cat([delta, torch.full((n_rays, 1), 1e10, device=dev)], dim=-1)
    else:
        delta = torch.cat([delta, torch.full((n_rays, 1), 1.0 / n_steps, device=dev)], dim=-1)
        last = F.one_hot(lengths - 1, S).bool()
        delta = torch.where(last, torch.full_like(delta, 1.0 / n_steps), delta)
    alpha = 1.0 - torch.exp(-dens * delta)
    trans = torch.cumprod(torch.cat([torch.ones(n_rays, 1, device=dev), 1.0 - alpha + 1e-7], dim=-1), dim=-1)[:, :-1]
    w = alpha * trans * slot
    return (rgb * w.unsqueeze(-1)).sum(dim=1), w.sum(dim=-1, keepdim=True).clip(0, 1)


# ------------------------------------------------------------------------------------------------
# input augmentation (models/__init__.py:157-174, utils/utils.py:183-230)
# ------------------------------------------------------------------------------------------------
def augm_rots(roll_range=90, pitch_range=90, yaw_range=90):
    """Random rotation about x, y, z (degrees; normal, uniform, normal draws clipped to +-2*range)."""
    def clipped(v, r):
        return min(2 * r, max(-2 * r, v))

    ax = np.deg2rad(clipped(np.random.randn() * roll_range, roll_range))
    ay = np.deg2rad(clipped(np.random.rand() * pitch_range, pitch_range))
    az = np.deg2rad(clipped(np.random.randn() * yaw_range, yaw_range))
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    return rx @ (ry @ rz)


# ------------------------------------------------------------------------------------------------
# loss (renderer/loss.py:6-191) and optimiser groups (lightning_model.py:403-461)
# ------------------------------------------------------------------------------------------------
class IDHRLoss(nn.Module):
    """rgb + perceptual + eikonal + mask + off-surface + inside + SDF-parameter + skinning terms, weighted."""

    TERMS = ("rgb", "perceptual", "eikonal", "mask", "off_surface", "inside", "sdf_params", "skinning")

    def __init__(self, rgb_weight, perceptual_weight, eikonal_weight, mask_weight, off_surface_weight, inside_weight,
                 params_weight, skinning_weight, rgb_loss_type="l1", perceptual_loss_fn=None):
        super().__init__()
        self.weights = dict(rgb=rgb_weight, perceptual=perceptual_weight, eikonal=eikonal_weight, mask=mask_weight,
                            off_surface=off_surface_weight, inside=inside_weight, sdf_params=params_weight,
                            skinning=skinning_weight)
        kinds = {"l1": nn.L1Loss(reduction="sum"), "mse": nn.MSELoss(reduction="sum"),
                 "smoothed_l1": nn.SmoothL1Loss(reduction="sum", beta=1e-1)}
        if rgb_loss_type not in kinds:
            raise ValueError("Unsupported RGB loss type: %s. Only l1, smoothed_l1 and mse are supported" % rgb_loss_type)
        self.pixel_loss = kinds[rgb_loss_type]
        self.p_loss = perceptual_loss_fn

    def forward(self, out, gt):
        dev = out["rgb_values"].device
        zero = lambda: 0.0     # (a term that is switched off: a number, not a launch)
        hit = out["network_body_mask"][:, :2048]
        body = out["body_mask"][:, :2048]
        off = out["off_surface_mask"][:, :2048]
        n_px = float(body.numel())
        terms = {k: zero() for k in self.TERMS}
        if self.weights["rgb"] > 0:
            # the reference compacts the hit pixels (boolean indexing: a device -> host round trip each) and sums the
            # element-wise loss over them; the same sum with the mask as a factor needs no round trip.  Patch sampling marks
            # boundary pixels with 100 (body.max() > 1 there); without such pixels the second factor is all ones.
            m = (hit & (body != 100)).unsqueeze(-1).to(out["rgb_values"].dtype)
            a, b = out["rgb_values"][:, :2048], gt["rgb"][:, :2048]
            if isinstance(self.pixel_loss, nn.L1Loss):
                per = (a - b).abs()
            elif isinstance(self.pixel_loss, nn.MSELoss):
                per = (a - b) ** 2
            else:
                per = F.smooth_l1_loss(a, b, reduction="none", beta=1e-1)
            terms["rgb"] = (per * m).sum() / float(hit.numel())
        if self.weights["perceptual"] > 0:
            if self.p_loss is None:
                raise ValueError("perceptual_weight > 0 needs a perceptual_loss_fn (e.g. LPIPS)")
            pred = out["rgb_values"][:, 2048:].reshape(-1, 48, 48, 3).permute(0, 3, 1, 2)
            ref = gt["rgb"][:, 2048:].reshape(-1, 48, 48, 3).permute(0, 3, 1, 2)
            terms["perceptual"] = self.p_loss(pred, ref, normalize=True).mean() if hit.sum() > 0 else torch.tensor(0.0, device=dev)
        if self.weights["mask"] > 0:
            # the L2 norm of the residual vector over the off-surface pixels (torch.norm over the last dim of a 1-D tensor);
            # masked-out entries contribute zeros, an empty mask gives 0 with a zero gradient -- no host round trip
            acc = out["sdf_output"]
            if acc.dim() == body.dim():
                terms["mask"] = torch.norm((acc[:, :off.shape[1]] - body.float()) * off) / n_px
            elif off.sum() == 0:
                terms["mask"] = torch.tensor(0.0, device=dev)
            else:   # a trailing singleton axis broadcasts against the mask values (n, 1) - (n,): the reference's arithmetic
                terms["mask"] = torch.norm(acc[off] - body[off].float(), dim=-1).sum() / n_px
        if self.weights["eikonal"] > 0:
            g = out["grad_theta"]
            terms["eikonal"] = (torch.abs(g.norm(2, dim=-1) - 1).sum() / n_px) if g.shape[0] else torch.tensor(0.0, device=dev)
        if self.weights["off_surface"] > 0:
            terms["off_surface"] = torch.exp(-1e2 * out["off_surface_sdf"]).sum() / n_px
        if self.weights["inside"] > 0:
            terms["inside"] = torch.sigmoid(out["inside_sdf"] * 5e3).sum() / n_px
        if self.weights["sdf_params"] > 0:
            p = torch.cat(out["sdf_params"], dim=1)
            terms["sdf_params"] = p.norm(dim=-1).mean() / p.size(-1)
        if self.weights["skinning"] > 0:
            terms["skinning"] = torch.abs(out["pred_weights"] - gt["sampled_weights"]).sum(-1).mean()
        # the weighted sum as ONE product with the weight vector (eight multiplications and seven additions were fifteen launches
        # and as many in backward); the terms in the reference's order
        live = [k for k in self.TERMS if torch.is_tensor(terms[k])]
        if live:
            vec = torch.stack([terms[k].reshape(()) for k in live])
            key = (str(vec.device), vec.dtype, tuple(live))
            if getattr(self, "_wvec_key", None) != key:     # the weights are constructor constants: on the device once
                self._wvec = torch.tensor([float(self.weights[k]) for k in live], dtype=vec.dtype, device=vec.device)
                self._wvec_key = key
            total = (vec * self._wvec).sum().reshape(1)     # (1,) like the reference's sum that starts from torch.zeros(1)
        else:
            total = torch.zeros(1, device=dev)
        res = {"loss": total}
        res.update({k + "_loss": (v if torch.is_tensor(v) else torch.zeros(1, device=dev)) for k, v in terms.items()})
        return res


def build_loss(cfg, perceptual_loss_fn=None):
    t = cfg["training"]
    return IDHRLoss(rgb_weight=t["rgb_weight"], perceptual_weight=t["perceptual_weight"],
                    eikonal_weight=t["eikonal_weight"], mask_weight=t["mask_weight"],
                    off_surface_weight=t["off_surface_weight"], inside_weight=t["inside_weight"],
                    params_weight=t["params_weight"], skinning_weight=t["skinning_weight"],
                    rgb_loss_type=t.get("rgb_loss_type", "l1"), perceptual_loss_fn=perceptual_loss_fn)


def configure_optimizers(model, cfg):
    """Adam with the reference's parameter groups, in the reference's ORDER (an optimiser state dict addresses groups by
    position) and with its learning rates (lightning_model.py:403-461)."""
    t, m = cfg["training"], cfg["model"]
    lr = t["lr"]
    groups = [{"params": model.sdf_decoder.net.layers.parameters(), "lr": lr},
              {"params": model.sdf_decoder.pose_encoder.parameters(), "lr": lr * t["pose_net_factor"]},
              {"params": model.color_decoder.parameters(), "lr": 1e-4},
              {"params": model.deviation_decoder.parameters(), "lr": 1e-4}]
    if t["train_skinning_net"]:
        groups.append({"params": model.skinning_model.parameters(), "lr": t["skinning_lr"]})
    if m.get("train_cameras"):
        groups.append({"params": model.camera_parameters(), "lr": 1e-4})                      # :434-440
    if m.get("train_smpl"):
        groups.append({"params": model.smpl_parameters(), "lr": 1e-4})                        # :442-448
    if m.get("color_pose_encoder") in ("hybrid", "latent") or m.get("geo_pose_encoder") in ("latent",):
        groups.append({"params": model.latent.parameters(), "lr": 1e-4, "weight_decay": 0.05})   # :450-457
    # fused: one multi-tensor kernel per group for the whole update instead of torch's foreach chain (~80 launches over the
    # 87 M parameters); same update rule, same state-dict keys.  CUDA parameters only.
    fused = all(p.is_cuda for p in model.parameters()) and os.environ.get("ARAH_FUSED_ADAM", "1") != "0"
    return torch.optim.Adam(params=groups, fused=True) if fused else torch.optim.Adam(params=groups)


def training_step(model, criteria, inputs):
    """compute_loss of the reference's harness (lightning_model.py:636-653): forward + loss dict."""
    out = model(inputs)
    gt = {"rgb": inputs["rgb_values"]}
    if "sampled_weights" in inputs:
        gt["sampled_weights"] = inputs["sampled_weights"]
    return criteria(out, gt)
