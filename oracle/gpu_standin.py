"""TEST / BENCH INFRASTRUCTURE, not product code: the stand-in for "the reference's single-GPU rays/s" (BASELINE.md section 2).

The reference's own GPU path cannot run on the benchmark box (it never travels; it needs CUDA-only pytorch3d), and it publishes no
number.  BASELINE.md names the stand-in: the reference's op sequence on PyTorch-ROCm.  This module runs oracle/arah_oracle.py -- the
CPU restatement that tests/test_oracle_golden.py pins against the reference's own outputs -- on `cuda` tensors, with what the
reference's GPU run does around those ops:

* `pytorch3d.ops.knn_points` (ray_tracing.py:386,407; CUDA brute force) -> an exact brute-force 1-NN on the device, in chunks
  (differences, not the |a|^2 - 2ab + |b|^2 expansion: exact like the reference's kernel; lowest index wins ties);
* `eval_sdf(..., eval_mode=True)` (root_finding_utils.py:116-144): batches of `point_batch_size` points, every batch's result
  moved to the host (`.detach().cpu()`) and the concatenation back to the device (:132-142);
* loop C in chunks of 1e6 points (root_finding_utils.py:316-318), loop D in chunks of 20 480 rays with 1e6-point batches
  (implicit_differentiable_renderer.py:194-198, 220);
* Broyden's active sets by boolean-mask indexing (a host synchronisation per iteration, broyden.py:64-65): as the oracle does.

Only bench.py's baseline legs and tests import this file; nothing under arah_release_amd/ does.
"""
import time

import torch

from . import arah_oracle as O

SDF_POINT_BATCH = 100000        # eval_sdf's default point_batch_size (root_finding_utils.py:116)
CANON_POINT_BATCH = 1000000     # search_canonical_corr's chunk (root_finding_utils.py:316-318)
KNN_CHUNK = 4096


def _nearest_vertex_device(fr, pts):
    fr.counters["n_knn"] += pts.shape[0]
    if pts.shape[0] == 0:
        return torch.zeros(0, dtype=torch.long, device=pts.device)
    out = []
    v = fr.verts
    for c in range(0, pts.shape[0], KNN_CHUNK):
        d2 = (pts[c:c + KNN_CHUNK, None, :] - v[None, :, :]).pow(2).sum(-1)
        out.append(d2.argmin(dim=1))
    return torch.cat(out)


def _sdf_forward_round_trip(fr, x, count=True):
    """eval_sdf(eval_mode=True): per batch the value goes to the host, the concatenation comes back (RFU:122-142).  The feature
    (second return value) is only used by loop D, which calls the network directly (IDR:336-337), not through eval_sdf."""
    if x.requires_grad or x.shape[0] == 0:
        return _SDF_FORWARD(fr, x, count)
    vals, feats = [], []
    for c in range(0, x.shape[0], SDF_POINT_BATCH):
        s, h = _SDF_FORWARD(fr, x[c:c + SDF_POINT_BATCH], count)
        vals.append(s.detach().cpu())
        feats.append(h)
    return torch.cat(vals).to(x.device), torch.cat(feats)


_SDF_FORWARD = O.sdf_forward
_CANONICALIZE = O.canonicalize


def _canonicalize_chunked(fr, pts):
    if pts.shape[0] <= CANON_POINT_BATCH:
        return _CANONICALIZE(fr, pts)
    parts = [_CANONICALIZE(fr, pts[c:c + CANON_POINT_BATCH]) for c in range(0, pts.shape[0], CANON_POINT_BATCH)]
    return tuple(torch.cat([p[i] for p in parts]) for i in range(3))


class _on_device:
    """The oracle's factory calls (torch.zeros, arange, linspace ...) make device tensors; its CPU-only pieces are swapped."""

    def __init__(self, device):
        self.ctx = torch.device(device)

    def __enter__(self):
        self.keep = (O.nearest_vertex, O.sdf_forward, O.canonicalize)
        O.nearest_vertex, O.sdf_forward, O.canonicalize = _nearest_vertex_device, _sdf_forward_round_trip, _canonicalize_chunked
        self.ctx.__enter__()

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        O.nearest_vertex, O.sdf_forward, O.canonicalize = self.keep


def render(model, inputs, cano_view_dirs, n_steps=64, n_near=16, n_far=16):
    """The oracle's whole eval forward on the device of `inputs` (a `cuda` model and input dict)."""
    dev = inputs["ray_dirs"].device
    with _on_device(dev), torch.no_grad():
        return O.render_inputs(model, inputs, cano_view_dirs, n_steps, n_near, n_far)


def timed(model, make_inputs, cano_view_dirs, n_rays, warm_rays, n_steps, n_near, n_far):
    """(result dict, seconds, rays) of one render of `n_rays` rays after a warm-up render of `warm_rays` rays
    (library initialisation, allocator pools)."""
    render(model, make_inputs(warm_rays), cano_view_dirs, n_steps, n_near, n_far)
    torch.cuda.synchronize()
    inputs = make_inputs(n_rays)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = render(model, inputs, cano_view_dirs, n_steps, n_near, n_far)
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0, int(inputs["ray_dirs"].shape[1])
