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
