"""Sample-axis products of the training step (arah_release_amd/tall.py), host side: the split-K gram, the linear layer
whose weight gradient goes through it, and the elementwise small matrix-vector product -- against plain torch."""
import torch

from arah_release_amd import tall


def test_gram_matches_matmul_for_ragged_and_sliced_operands():
    g = torch.Generator().manual_seed(1)
    for P in (1, 63, 4096, 4097, 70001):
        wide_a = torch.randn(P, 40, generator=g, dtype=torch.float64)
        wide_b = torch.randn(P, 50, generator=g, dtype=torch.float64)
        for a, b in ((wide_a, wide_b), (wide_a[:, :33], wide_b[:, 3:20]), (wide_a[:, :1], wide_b), (wide_a, wide_b[:, :3])):
            got = tall.gram(a, b)
            assert got.shape == (a.shape[1], b.shape[1])
            torch.testing.assert_close(got, a.t() @ b, rtol=1e-12, atol=1e-9)


def test_tall_linear_forward_and_gradients():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(5000, 24, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(7, 24, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(7, generator=g, dtype=torch.float64, requires_grad=True)
    up = torch.randn(5000, 7, generator=g, dtype=torch.float64)
    y = tall.tall_linear(x, w, b)
    ref = torch.nn.functional.linear(x, w, b)
    torch.testing.assert_close(y, ref)
    got = torch.autograd.grad(y, (x, w, b), up)
    want = torch.autograd.grad(ref, (x, w, b), up)
    for a, c in zip(got, want):
        torch.testing.assert_close(a, c, rtol=1e-12, atol=1e-10)
    assert torch.autograd.gradcheck(lambda *a: tall._TallLinear.apply(*a), (x[:50], w, b))
    # small inputs and no-grad calls take the plain path
    with torch.no_grad():
        torch.testing.assert_close(tall.tall_linear(x, w, b), ref)
    torch.testing.assert_close(tall.tall_linear(x[:10], w, b), ref[:10])


def test_mv3_is_a_batched_matrix_vector_product():
    g = torch.Generator().manual_seed(3)
    M = torch.randn(2, 9, 3, 3, generator=g)
    v = torch.randn(2, 9, 3, generator=g)
    torch.testing.assert_close(tall.mv3(M, v), torch.matmul(M, v.unsqueeze(-1)).squeeze(-1))
    M34 = torch.randn(11, 3, 4, generator=g)
    v4 = torch.randn(11, 4, generator=g)
    torch.testing.assert_close(tall.mv3(M34, v4), torch.einsum("pij,pj->pi", M34, v4))


def test_gram_grouped_is_a_batched_transposed_product():
    """tall.gram_grouped (round 6: G weight-gradient products over the sample axis as one batched split-K launch on grouped,
    zero-padded operand streams) against a float64 bmm; the split path, the plain path for short operands, padding rows."""
    g = torch.Generator().manual_seed(4)
    for G, R, m, n in ((3, 64 * 20, 8, 5), (2, 64 * 16, 16, 16), (1, 640, 4, 3), (4, 100, 3, 2)):
        a = torch.randn(G, R, m, generator=g, dtype=torch.float64)
        b = torch.randn(G, R, n, generator=g, dtype=torch.float64)
        a[:, R - 7:] = 0.0                                          # padding rows: zero in one operand is enough
        ref = torch.bmm(a.transpose(1, 2), b)
        torch.testing.assert_close(tall.gram_grouped(a, b), ref, rtol=1e-12, atol=1e-10)
        torch.testing.assert_close(tall.gram_grouped(a, b)[0], tall.gram(a[0], b[0]), rtol=1e-12, atol=1e-10)


def test_fold_cache_folds_a_weight_normed_layer_once():
    """nets.fold_cache: inside the context a weight-normed layer evaluated several times is folded once (one autograd node), the
    values and the gradients are those of the repeated folding."""
    from arah_release_amd import nets
    torch.manual_seed(0)
    lin = nets._wn_linear(6, 4, True)
    x1, x2 = torch.randn(5, 6), torch.randn(7, 6)

    def run(cached):
        lin.zero_grad(set_to_none=True)
        ctx = nets.fold_cache() if cached else __import__("contextlib").nullcontext()
        with ctx:
            w1, w2 = nets.folded_weight(lin), nets.folded_weight(lin)
            y = (x1 @ w1.t()).sum() + (x2 @ w2.t()).pow(2).sum()
        y.backward()
        return (w1 is w2), y.detach().clone(), lin.weight_g.grad.clone(), lin.weight_v.grad.clone()

    same0, y0, gg0, gv0 = run(False)
    same1, y1, gg1, gv1 = run(True)
    assert not same0 and same1
    torch.testing.assert_close(y1, y0)
    torch.testing.assert_close(gg1, gg0)
    torch.testing.assert_close(gv1, gv0)
    with torch.no_grad(), nets.fold_cache():                         # no graph: nothing is cached across calls
        assert nets.folded_weight(lin) is not nets.folded_weight(lin)
