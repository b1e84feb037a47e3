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
