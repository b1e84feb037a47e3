"""Products over the SAMPLE axis of a training step (P ~ 1e5 samples, feature widths <= 304).

The GEMM library sizes its tiles for square problems: a^T b with K = P and m, n <= 304 gets <= 32 workgroups for 256
CUs (357 us for 256 x 256 x 112 k), P batched 3 x 3 matrix-vector products take > 1 ms as a batched GEMM, and with
m = 1 the library's kernel runs at 1 % of HBM speed (profiles/r02_train_step_*).  The helpers below keep those
products on the shapes the machine is good at: a batched split-K product, one pass at HBM speed for the skinny ones
(arah_gram_skinny), plain elementwise arithmetic for the 3 x 3 ones."""
import torch
import torch.nn.functional as F


def gram(a, b, chunks=64):
    """a^T b for tall operands a (P, m), b (P, n); column slices of wider tensors are fine."""
    P = a.shape[0]
    per = P // chunks
    if a.is_cuda and P >= 4096 and min(a.shape[1], b.shape[1]) <= 4 and a.dtype == torch.float32:
        from . import hip   # a head or a K = 3 first layer: one pass over the wide operand
        return hip.gram_skinny(a, b) if a.shape[1] <= 4 else hip.gram_skinny(b, a).t()
    if per < 64:
        return a.t() @ b
    main = per * chunks
    out = torch.bmm(a[:main].reshape(chunks, per, a.shape[1]).transpose(1, 2),
                    b[:main].reshape(chunks, per, b.shape[1])).sum(0)
    if main < P:
        out = out + a[main:].t() @ b[main:]
    return out


def gram_grouped(a, b, chunks=64):
    """G products over the sample axis in one batched split-K launch: a (G, R, m), b (G, R, n) contiguous with R a multiple of
    `chunks` (rows that are padding must be zero in one operand: hip._grouped) -> (G, m, n).  The operands are only viewed."""
    G, R, m = a.shape
    n = b.shape[2]
    per = R // chunks
    if per < 16 or R % chunks:
        return torch.bmm(a.transpose(1, 2), b)
    out = torch.bmm(a.reshape(G * chunks, per, m).transpose(1, 2), b.reshape(G * chunks, per, n))
    return out.reshape(G, chunks, m, n).sum(1)


class _TallLinear(torch.autograd.Function):
    """x W^T + b for x (P, in); the weight gradient is a `gram` over the sample axis."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = gram(g, x) if ctx.needs_input_grad[1] else None
        gb = None
        if ctx.needs_input_grad[2]:
            if g.is_cuda and g.dtype == torch.float32:
                from . import hip   # one pass at HBM speed whatever the width (torch's reduction: 0.3 ms for a 25-wide g)
                gb = hip.colsum(g)
            else:
                gb = g.sum(0)
        return gx, gw, gb


def tall_linear(x, weight, bias):
    """F.linear for a (P, in) operand; first-order autograd only (the skinning MLP is never differentiated twice)."""
    if x.dim() == 2 and x.shape[0] >= 4096 and torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad):
        return _TallLinear.apply(x, weight, bias)
    return F.linear(x, weight, bias)


def mv3(M, v):
    """Batched small matrix-vector product M (..., r, c) v (..., c) -> (..., r) as elementwise work."""
    return (M * v.unsqueeze(-2)).sum(-1)
