"""Training path, multi-process: data-parallel over views with a gradient all-reduce.
Runs on CPU with the gloo backend and world_size 2.  The ray tracer (loops A-C, HIP only) is replaced by the
reference's own tracer output from fixture f5, so what is exercised is everything that carries gradients
(training.py) plus torch DDP: the all-reduced gradient of every one of the 211 parameters must equal the mean
of the two ranks' local gradients."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub_tracer(g, sel):
    T34 = torch.from_numpy(g["sampler_transforms34"][sel]).reshape(len(sel), -1, 3, 4)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(T34.shape[0], T34.shape[1], 1, 4)
    T44 = torch.cat([T34, bottom], dim=2)
    tup = (torch.from_numpy(g["points_hat_norm"][sel])[None], torch.from_numpy(g["network_body_mask"][sel])[None],
           torch.from_numpy(g["dists"][sel])[None], torch.from_numpy(g["sampler_pts"][sel])[None],
           torch.from_numpy(g["sampler_dists"][sel])[None], T44[None],
           torch.from_numpy(g["sampler_converge_mask"][sel])[None])
    return lambda *a, **k: tup


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arah_release_amd import config, synthetic, training
    g = golden("f5_tracer_s64.npz")
    model, cfg = config.build_synthetic_model("zju377_mono", training=dict(pose_input_noise=False, view_input_noise=False))
    model.train()
    scene = synthetic.SyntheticScene(0)
    full = scene.make_inputs(int(g["H"]), int(g["W"]), frame_idx=int(g["frame_idx"]), max_rays=int(g["max_rays"]),
                             eval_mode=False)
    sel = np.arange(rank, 256, world)            # each rank renders its own rays (a different "view")
    inputs = dict(full)
    for k in ("ray_dirs", "body_bounds_intersections", "body_mask", "rgb_values"):
        inputs[k] = full[k][:, sel]
    inputs["pose_cond"] = dict(full["pose_cond"])
    model.idhr_network.ray_tracer.forward = _stub_tracer(g, sel)
    crit = training.build_loss(cfg)
    torch.manual_seed(100 + rank)                # eikonal points differ per rank, like independent workers
    rng_state = torch.random.get_rng_state()
    training.training_step(model, crit, dict(inputs, pose_cond=dict(inputs["pose_cond"])))["loss"].backward()
    local = {n: p.grad.clone() for n, p in model.named_parameters()}
    assert len(local) == 211
    model.zero_grad(set_to_none=True)
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    torch.random.set_rng_state(rng_state)
    training.training_step(ddp, crit, dict(inputs, pose_cond=dict(inputs["pose_cond"])))["loss"].backward()
    worst = 0.0
    for n, p in model.named_parameters():
        mean = local[n].clone()
        dist.all_reduce(mean)
        mean /= world
        denom = float(mean.abs().max()) + 1e-12
        worst = max(worst, float((p.grad - mean).abs().max()) / denom)
    out[rank] = worst
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gradient_allreduce_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == world
    for r in range(world):
        assert out[r] < 1e-5, out[r]
