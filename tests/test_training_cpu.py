"""Host-side pieces of the training step against the reference (CPU)."""
import numpy as np
import pytest
import torch

from conftest import golden


@pytest.mark.parametrize("case", range(5))
def test_idhr_loss_against_reference(case):
    """Every term of IDHRLoss (reference renderer/loss.py) on the random model outputs of fixture F13: three pixel-loss
    types, boundary value 100 in the body mask, empty hit / off-surface masks, rays beyond the first 2048."""
    from arah_release_amd import training
    g = golden("f13_idhr_loss.npz")
    pre = "c%d." % case
    T = lambda a: torch.from_numpy(np.asarray(a))
    out = {k[len(pre) + 4:]: T(g[k]) for k in g.files if k.startswith(pre + "out.") and "sdf_params_" not in k}
    out["sdf_params"] = [T(g[pre + "out.sdf_params_%d" % k]) for k in range(int(g[pre + "n_sdf_params"]))]
    gt = {k[len(pre) + 3:]: T(g[k]) for k in g.files if k.startswith(pre + "gt.")}
    crit = training.IDHRLoss(1.0, 0.0, 0.1, 0.5, 0.3, 0.2, 1e-3, 10.0, rgb_loss_type=str(g[pre + "kind"]))
    res = crit(out, gt)
    names = [k[len(pre) + 4:] for k in g.files if k.startswith(pre + "res.")]
    assert set(names) == set(res), (sorted(names), sorted(res))
    for k in names:
        np.testing.assert_allclose(np.asarray(res[k].detach().numpy(), np.float64).reshape(-1), g[pre + "res." + k].reshape(-1),
                                   rtol=1e-6, atol=1e-9, err_msg=k)
    with pytest.raises(ValueError):
        training.IDHRLoss(1, 0, 0, 0, 0, 0, 0, 0, rgb_loss_type="huber")


def test_hierarchical_softmax_is_the_recursion():
    """training.hierarchical_softmax gathers each weight's factors and multiplies them in the order the reference's
    recursion does (utils/utils.py:138-181): the golden vectors of F2 bit for bit, and the same gradient."""
    from arah_release_amd import training
    from oracle import arah_oracle as O
    g = golden("f2_pointwise.npz")
    x = torch.from_numpy(g["logits"])
    w = training.hierarchical_softmax(x)
    np.testing.assert_allclose(w.numpy(), g["hsoftmax"], rtol=1e-6, atol=1e-7)
    assert torch.equal(w, O.hierarchical_softmax(x))
    xs = (torch.randn(257, 25, generator=torch.Generator().manual_seed(3)) * 4).requires_grad_(True)
    ga = torch.autograd.grad((training.hierarchical_softmax(xs) ** 2).sum(), xs)[0]
    gb = torch.autograd.grad((O.hierarchical_softmax(xs) ** 2).sum(), xs)[0]
    np.testing.assert_allclose(ga.numpy(), gb.numpy(), rtol=1e-5, atol=1e-6)
    paths = training._hsoftmax_paths()
    assert len(paths) == 24 and len({len(p) for p in paths}) == 1


@pytest.mark.parametrize("rel", [False, True])
def test_pose_encoder_levels_against_joint_by_joint(rel):
    """The level-batched HierarchicalPoseEncoder (training) against the joint-by-joint form (siren_modules.py:196-244):
    outputs and every parameter gradient; inference (no_grad) stays on the joint-by-joint order."""
    from arah_release_amd import nets
    torch.manual_seed(1)
    enc = nets.HierarchicalPoseEncoder(rel_joints=rel)
    rots, J = torch.randn(2, 24, 9), torch.randn(2, 24, 3)
    enc.batched_levels = True
    a = enc(rots, J)
    ga = torch.autograd.grad((a ** 2).sum(), list(enc.parameters()))
    with torch.no_grad():
        a_inf = enc(rots, J)
    enc.batched_levels = False
    b = enc(rots, J)
    gb = torch.autograd.grad((b ** 2).sum(), list(enc.parameters()))
    assert torch.equal(a_inf, b)
    np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), rtol=1e-5, atol=1e-6)
    for x, y in zip(ga, gb):
        np.testing.assert_allclose(x.numpy(), y.numpy(), rtol=1e-4, atol=1e-5)
    assert [len(l) for l in enc._levels()] == [1, 3, 3, 3, 5, 3, 2, 2, 2]


def test_hypernetwork_heads_batched_against_head_by_head():
    """HyperFCFiLM with the two normed layers of its seven heads as batched products (training) against the head-by-head
    modules (hyperlayers.py:270-285): emitted parameters and every parameter gradient; inference stays head by head."""
    from arah_release_amd import nets
    torch.manual_seed(0)
    net = nets.HyperBVPNet(out_features=1, in_features=3, hyper_in_ch=144, hidden_features=256, num_hidden_layers=5,
                           hierarchical_pose=True, use_FiLM=True)
    for l in net.net.layers:   # the residual heads start at zero: give them something to emit
        h = l.hyper_linear if hasattr(l, "hyper_linear") else l
        torch.nn.init.normal_(h.hypo_params.net[2].weight, std=1e-3)
    rots, J, lat = torch.randn(1, 24, 9), torch.randn(1, 24, 3), torch.randn(1, 128)

    def run(flag):
        net.net.batched_heads = flag
        out = net({"coords": torch.zeros(1, 1, 3), "rots": rots, "Jtrs": J, "latent": lat})
        params = torch.cat([p.reshape(-1) for p in out["params"]])
        grads = torch.autograd.grad((params ** 2).sum() * 1e3, list(net.parameters()), allow_unused=True)
        return params.detach(), grads

    a, ga = run(True)
    b, gb = run(False)
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-7)
    for x, y in zip(ga, gb):
        assert (x is None) == (y is None)
        if x is not None:
            np.testing.assert_allclose(x.numpy(), y.numpy(), rtol=1e-4, atol=1e-6 * float(y.abs().max()) + 1e-12)
    outs = []
    for flag in (True, False):   # inference ignores the switch
        net.net.batched_heads = flag
        with torch.no_grad():
            o = net({"coords": torch.zeros(1, 1, 3), "rots": rots, "Jtrs": J, "latent": lat})
        outs.append(torch.cat([p.reshape(-1) for p in o["params"]]))
    assert torch.equal(outs[0], outs[1])
