"""Fit the synthetic subject's networks once and store them as a committed asset.

No checkpoints of the reference are redistributable (they live on Google Drive and depend on
licence-gated SMPL data), so parity tests and benchmarks use seeded synthetic weights:

  * SDF MLP   : FiLM-SIREN 3->256x6->1 fitted (Adam) to the capsule-union SDF of the synthetic body
                in normalised canonical coordinates, with fixed non-trivial FiLM vectors;
  * skin MLP  : Deformer 3->128x4->25 fitted so that hierarchical_softmax(20*logits) reproduces the
                synthetic body's soft skinning weights (so forward-LBS root finding behaves like on
                a real subject);
  * colour MLP: default-initialised weight-normed MLPs for the two reference modes
                ('no_view_dir' 390-in of ZJUMOCAP-377-mono, 'idr' 417-in of ZJUMOCAP-313);
  * latent    : 4 x 128 embedding, N(0, 1).

Output: arah_release_amd/assets/synthetic_weights.npz (fp32).  Deterministic given the seeds
below; re-running it is only needed if the synthetic body changes.

    python tools/make_synthetic_assets.py [--sdf-steps 1500] [--skin-steps 2000]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from arah_release_amd import nets, synthetic  # noqa: E402


def siren_init(n_in, n_out, first):
    w = torch.empty(n_out, n_in)
    if first:
        w.uniform_(-1.0 / n_in, 1.0 / n_in)
    else:
        lim = np.sqrt(6.0 / n_in) / 30.0
        w.uniform_(-lim, lim)
    b = torch.empty(n_out).uniform_(-1.0 / np.sqrt(n_in), 1.0 / np.sqrt(n_in))
    return w.requires_grad_(True), b.requires_grad_(True)


def hsoftmax(x):
    """Differentiable torch restatement used only for fitting (semantics: utils/utils.py:138-181)."""
    sg = torch.sigmoid(x)
    w = [None] * 24
    s0 = torch.softmax(x[:, 1:4], dim=-1)
    w[0] = 1 - sg[:, 0]
    for k in range(3):
        w[1 + k] = sg[:, 0] * s0[:, k]
    for p, c in ((1, 4), (2, 5), (3, 6), (4, 7), (5, 8), (6, 9), (7, 10), (8, 11)):
        w[c] = w[p] * sg[:, c]
        w[p] = w[p] * (1 - sg[:, c])
    s1 = torch.softmax(x[:, 12:15], dim=-1)
    for k in range(3):
        w[12 + k] = w[9] * sg[:, 24] * s1[:, k]
    w[9] = w[9] * (1 - sg[:, 24])
    for p, c in ((12, 15), (13, 16), (14, 17), (16, 18), (17, 19), (18, 20), (19, 21), (20, 22), (21, 23)):
        w[c] = w[p] * sg[:, c]
        w[p] = w[p] * (1 - sg[:, c])
    return torch.stack(w, dim=-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sdf-steps", type=int, default=1500)
    ap.add_argument("--skin-steps", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=8192)
    args = ap.parse_args()
    torch.manual_seed(20240926)
    rng = np.random.RandomState(7)
    scene = synthetic.SyntheticScene(seed=0)
    cmin, cmax, center = float(scene.coord_min), float(scene.coord_max), scene.center
    scale = (cmax - cmin) * 1.1 / 2.0
    unnorm = lambda xn: (xn / 2.0 + 0.5) * 1.1 * (cmax - cmin) + cmin - (cmax - cmin) * 0.05 + center

    def sample_points(n):
        k = n // 2
        sv = scene.verts_cano[rng.randint(0, synthetic.N_VERTS, k)] + rng.randn(k, 3) * 0.03
        near = synthetic.normalize_points_np(sv, cmin, cmax, center)
        uni = rng.rand(n - k, 3) * 2 - 1
        return np.concatenate([near, uni], 0).astype(np.float32)

    # ---------------- SDF SIREN with fixed FiLM vectors
    freq = (1.0 + 0.05 * torch.randn(6, 256)).float()
    phase = (0.05 * torch.randn(6, 256)).float()
    dims = [3] + [256] * 6 + [1]
    params = [siren_init(dims[i], dims[i + 1], i == 0) for i in range(7)]
    opt = torch.optim.Adam([p for wb in params for p in wb], lr=1e-4)

    def siren(x):
        h = x
        for i in range(6):
            h = torch.sin(30.0 * (freq[i] * (h @ params[i][0].t() + params[i][1]) + phase[i]))
        return (h @ params[6][0].t() + params[6][1])[:, 0]

    t0 = time.time()
    for it in range(args.sdf_steps):
        xn = sample_points(args.batch)
        tgt = torch.from_numpy((synthetic.capsule_union_sdf(unnorm(xn)) / scale).astype(np.float32))
        x = torch.from_numpy(xn).requires_grad_(True)
        pred = siren(x)
        grad = torch.autograd.grad(pred.sum(), x, create_graph=True)[0]
        loss_sdf = (pred - tgt).abs().mean()
        loss_eik = ((grad.norm(dim=-1) - 1.0) ** 2).mean()
        loss = loss_sdf + 0.02 * loss_eik
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it % 100 == 0 or it == args.sdf_steps - 1:
            print("sdf  it %5d  |err| %.5f (%.2f mm)  eik %.4f  %.0fs" %
                  (it, loss_sdf.item(), loss_sdf.item() * scale * 1e3, loss_eik.item(), time.time() - t0), flush=True)

    # ---------------- skinning MLP
    skin = nets.Deformer(d_in=3, d_out=25, d_hidden=128, n_layers=4, skip_in=[], cond_in=[], multires=0,
                         bias=1.0, geometric_init=False, weight_norm=True)
    opt = torch.optim.Adam(skin.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(args.skin_steps // 3, 1), gamma=0.3)
    t0 = time.time()
    for it in range(args.skin_steps):
        xn = sample_points(args.batch)
        tgt = torch.from_numpy(synthetic.soft_skinning_weights(unnorm(xn)).astype(np.float32))
        pred = hsoftmax(skin(torch.from_numpy(xn)[None])[0] * 20.0)
        loss = (pred - tgt).abs().sum(-1).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if it % 200 == 0 or it == args.skin_steps - 1:
            print("skin it %5d  L1 %.4f  %.0fs" % (it, loss.item(), time.time() - t0), flush=True)

    out = {"film_freq": freq.numpy(), "film_phase": phase.numpy()}
    for i, (w, b) in enumerate(params):
        out["sdf_w%d" % i] = w.detach().numpy()
        out["sdf_b%d" % i] = b.detach().numpy()
    for k, v in skin.state_dict().items():
        out["skin." + k] = v.numpy()

    # ---------------- colour MLPs (both reference modes) + latent codes + beta
    for tag, kw in (("no_view_dir", dict(mode="no_view_dir", d_in=6, multires_view=0)),
                    ("idr", dict(mode="idr", d_in=9, multires_view=4))):
        torch.manual_seed(11 if tag == "idr" else 12)
        col = nets.RenderingNetwork(d_feature=256 + 128, d_out=3, d_hidden=256, n_layers=5, weight_norm=True,
                                    multires=0, skips=[3], squeeze_out=True, pose_encoder="latent", **kw)
        with torch.no_grad():  # make colours vary visibly with position/normal
            col.lin0.weight_v[:, :6] *= 4.0
            col.lin5.weight_g *= 3.0
        for k, v in col.state_dict().items():
            out["color_%s.%s" % (tag, k)] = v.numpy()
    torch.manual_seed(13)
    out["latent"] = torch.randn(4, 128).numpy()
    out["variance"] = np.float32(1e-3)

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "arah_release_amd", "assets")
    os.makedirs(dst, exist_ok=True)
    path = os.path.join(dst, "synthetic_weights.npz")
    np.savez(path, **{k: np.asarray(v, dtype=np.float32) for k, v in out.items()})
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
