"""bench.py's baseline legs on the CPU: the PyTorch stand-in of the reference's GPU path is the oracle's op sequence (same image
when run on the CPU device with small batch sizes), and the multi-process CPU baseline renders the same rays as the single
process."""
import os
import sys

import numpy as np
import torch

from conftest import REPO, get_model


def test_gpu_standin_is_the_oracles_op_sequence(scene):
    """oracle/gpu_standin.py swaps the oracle's CPU-only pieces (cKDTree -> brute-force 1-NN on the device, eval_sdf's host
    round trips, 1e6-point chunks) and runs the same functions: on the CPU device, with batch sizes small enough to exercise
    every chunk boundary, it must give the oracle's image bit for bit."""
    from oracle import arah_oracle as O, gpu_standin as G
    model, cfg = get_model("zju377_mono")
    cvd = cfg["model"]["cano_view_dirs"]
    a = O.render_inputs(model, scene.make_inputs(64, 64, frame_idx=0, max_rays=200), cvd, 64, 16, 16)
    keep = (G.SDF_POINT_BATCH, G.CANON_POINT_BATCH, G.KNN_CHUNK)
    G.SDF_POINT_BATCH, G.CANON_POINT_BATCH, G.KNN_CHUNK = 999, 4001, 257
    try:
        b = G.render(model, scene.make_inputs(64, 64, frame_idx=0, max_rays=200), cvd, 64, 16, 16)
    finally:
        G.SDF_POINT_BATCH, G.CANON_POINT_BATCH, G.KNN_CHUNK = keep
    assert O.nearest_vertex.__name__ == "nearest_vertex" and O.sdf_forward.__name__ == "sdf_forward"   # swapped back
    assert torch.equal(a["rgb_values"], b["rgb_values"]) and torch.equal(a["network_body_mask"], b["network_body_mask"])
    assert a["frame"].counters == b["frame"].counters


def test_ray_slices_partition_the_inputs(scene):
    sys.path.insert(0, REPO)
    import bench
    inputs = scene.make_inputs(64, 64, frame_idx=0, max_rays=100)
    n = inputs["ray_dirs"].shape[1]
    parts = [bench._slice_rays(inputs, lo, min(n, lo + 37)) for lo in range(0, n, 37)]
    assert sum(p["ray_dirs"].shape[1] for p in parts) == n
    assert torch.equal(torch.cat([p["ray_dirs"] for p in parts], dim=1), inputs["ray_dirs"])
    assert torch.equal(torch.cat([p["body_bounds_intersections"] for p in parts], dim=1), inputs["body_bounds_intersections"])
    assert parts[0]["smpl_verts"] is inputs["smpl_verts"]
