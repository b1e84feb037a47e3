"""The native train entry point (arah_release_amd/train.py; reference train.py:1-140 + what Lightning did for it)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from arah_release_amd import train


def test_epoch_indices_are_the_distributed_samplers():
    from torch.utils.data import DistributedSampler
    for n, world in ((5, 2), (8, 4), (7, 1), (3, 4)):
        for epoch in (0, 3):
            seen = []
            for rank in range(world):
                s = DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=True, seed=0)
                s.set_epoch(epoch)
                got = train.epoch_indices(n, epoch, rank, world)
                if n >= world:            # (torch pads differently when the dataset is smaller than the world)
                    assert got == list(iter(s)), (n, world, epoch, rank)
                seen += got
            assert set(seen) == set(range(n))


def test_epochs_to_run_chains_jobs_like_the_reference():
    assert train.epochs_to_run(250, -1, None) == 250 and train.epochs_to_run(250, -1, 40) == 250      # train.py:107-110
    assert train.epochs_to_run(250, 10, None) == 10                                                    # :115-116
    assert train.epochs_to_run(250, 10, 40) == 50 and train.epochs_to_run(250, 10, 245) == 250         # :118-119


def test_checkpoint_layout_round_trip(tmp_path):
    """Lightning's layout: 'epoch', 'global_step', 'state_dict' with the 'model.' prefix, 'optimizer_states' -- what
    get_model(checkpoint_path=...) of this build and of the reference read (metaavatar_render/config.py:253-300)."""
    from arah_release_amd import config
    cfg = config.builtin_config("zju313")
    lm = config.get_model(cfg, mode="test", n_data_points=3)
    with torch.no_grad():
        lm.model.deviation_decoder.variance.fill_(0.0123)
    opt = lm.configure_optimizers()
    path = str(tmp_path / "checkpoints" / "last.ckpt")
    train.save_checkpoint(path, lm, opt, epoch=7, global_step=99)
    ck = torch.load(path, map_location="cpu")
    assert ck["epoch"] == 7 and ck["global_step"] == 99 and len(ck["optimizer_states"]) == 1
    assert all(k.startswith("model.") for k in ck["state_dict"]) and ck["state_dict"]["model.latent.weight"].shape[0] == 3
    again = config.get_model(cfg, mode="test", checkpoint_path=path)
    assert float(again.model.deviation_decoder.variance) == pytest.approx(0.0123)
    assert again.model.latent.num_embeddings == 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reduce_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(rank)
    params = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2))]
    params[0].grad = torch.full((3, 4), float(rank + 1))
    params[1].grad = torch.arange(5.0) * (rank + 1)
    train.allreduce_gradients(params, world, dist)          # params[2] has no gradient: skipped on every rank alike
    out[rank] = (params[0].grad.clone(), params[1].grad.clone(), params[2].grad)
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_gradients_two_ranks_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_reduce_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in range(2):
        g0, g1, g2 = out[r]
        assert torch.equal(g0, torch.full((3, 4), 1.5)) and torch.equal(g1, torch.arange(5.0) * 1.5) and g2 is None


def _exchange_worker(rank, world, port, out, backend="gloo"):
    """Two replicas whose items touch DIFFERENT parameters (per-frame parameters under train_smpl): gradients arrive through
    the hooks during backward(), small buckets force several collectives, one parameter is used by no rank.
    backend "nccl" (= RCCL): one GPU per rank, the exchange on its own stream behind events."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cpu")
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    elif backend == "gloo_cuda":   # device tensors, both ranks on GPU 0, gloo moving them: the stream / event side of the exchange
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                              # replicas start from different draws ...
    net = torch.nn.ModuleDict({"shared": torch.nn.Linear(6, 5), "frames": torch.nn.ParameterDict(
        {"pose_0": torch.nn.Parameter(torch.randn(6)), "pose_1": torch.nn.Parameter(torch.randn(6)),
         "pose_2": torch.nn.Parameter(torch.randn(6))}), "head": torch.nn.Linear(5, 1)}).to(dev)
    train.broadcast_state(net, world, dist)                    # ... and are made equal, like Lightning's DDP does
    start = {k: v.clone() for k, v in net.state_dict().items()}
    params = list(net.parameters())
    ex = train.GradientExchange(params, world, dist, bucket_bytes=64)
    assert len(ex.buckets) > 2
    opt = torch.optim.Adam(params, lr=0.1)
    stops = []
    for step in range(3):
        opt.zero_grad(set_to_none=True)
        x = net["frames"]["pose_%d" % rank] * (step + 1.0)     # rank r draws frame r: pose_2 is never used
        loss = net["head"](torch.tanh(net["shared"](x))).sum()
        loss.backward()
        stops.append(ex.finish(stop=(rank == 1 and step == 2)))
        if step == 0:
            grads = {n: (None if p.grad is None else p.grad.detach().cpu().clone()) for n, p in net.named_parameters()}
        opt.step()
    out[rank] = ({k: v.cpu() for k, v in start.items()}, grads, {k: v.cpu().clone() for k, v in net.state_dict().items()}, stops,
                 loss.item())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_gradient_exchange_with_rank_dependent_parameter_sets_rccl():
    """The same exchange over RCCL, one GPU per rank: runs wherever two GPUs are visible (the driver's scaling node), skips
    on a one-GPU box.  tools/scale.sh launches the full-size counterparts (bench.py --gpus N, train_bench.py --exchange native)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _check_exchange("nccl", rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_gradient_exchange_on_device_tensors():
    """The exchange with its parameters on the GPU (one GPU shared by both ranks, gloo as the transport): the collectives are
    issued from the exchange's own stream behind events recorded after the bucket copies."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _check_exchange("gloo_cuda", rtol=1e-5, atol=1e-6)


def test_gradient_exchange_with_rank_dependent_parameter_sets():
    _check_exchange("gloo", rtol=1e-6, atol=1e-7)


def _check_exchange(backend, rtol, atol):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_exchange_worker, args=(2, _free_port(), out, backend), nprocs=2, join=True)
    (s0, g0, e0, stop0, _), (s1, g1, e1, stop1, _) = out[0], out[1]
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k                    # broadcast_state
    assert g0["frames.pose_2"] is None and g1["frames.pose_2"] is None            # used by no rank: stays None
    for k in g0:
        if g0[k] is not None:
            assert torch.equal(g0[k], g1[k]), k                # every rank holds the same averaged gradient
    # reference: the mean over ranks of the local gradients, absent contributions counted as zero
    torch.manual_seed(0)
    net = torch.nn.ModuleDict({"shared": torch.nn.Linear(6, 5), "frames": torch.nn.ParameterDict(
        {"pose_0": torch.nn.Parameter(torch.zeros(6)), "pose_1": torch.nn.Parameter(torch.zeros(6)),
         "pose_2": torch.nn.Parameter(torch.zeros(6))}), "head": torch.nn.Linear(5, 1)})
    net.load_state_dict(s0)
    want = {n: torch.zeros_like(p) for n, p in net.named_parameters()}
    for rank in range(2):
        net.zero_grad(set_to_none=True)
        net["head"](torch.tanh(net["shared"](net["frames"]["pose_%d" % rank]))).sum().backward()
        for n, p in net.named_parameters():
            if p.grad is not None:
                want[n] += p.grad / 2
    for n in want:
        if g0[n] is not None:
            torch.testing.assert_close(g0[n], want[n], rtol=rtol, atol=atol)
    assert not torch.equal(e0["frames.pose_0"], s0["frames.pose_0"]) and torch.equal(e0["frames.pose_2"], s0["frames.pose_2"])
    for k in e0:
        assert torch.equal(e0[k], e1[k]), k                    # replicas stay identical after three Adam steps
    assert stop0 == stop1 == [False, False, True]              # the stop request of one rank reaches every rank, same step


@pytest.mark.gpu
def test_train_entry_end_to_end(tmp_path, scene, monkeypatch):
    """python -m arah_release_amd.train on a capture written in the reference's layout: two epochs over two views, the
    checkpoint in Lightning's layout, resumed by a second run with --epochs-per-run, loaded by the test entry point."""
    import json
    import yaml
    from PIL import Image
    from arah_release_amd import config, data, smpl
    body = smpl.BodyModel.synthetic(scene)
    sub = tmp_path / "data" / "CoreView_000"
    (sub / "models").mkdir(parents=True)
    (sub / "1").mkdir()
    H = W = 256
    K = [[300.0, 0, 128], [0, 300.0, 128], [0, 0, 1]]
    (sub / "cam_params.json").write_text(json.dumps({"all_cam_names": ["1"], "1": {"K": K, "D": [0.0] * 5, "R": np.eye(3).tolist(),
                                                                                  "T": [[0], [0], [0.2]]}}))
    rng = np.random.RandomState(0)
    for f in range(2):
        fr = scene.frame(f)
        np.savez(sub / "models" / ("%06d.npz" % f), minimal_shape=scene.verts_cano, betas=np.zeros((1, 10), np.float32),
                 Jtr_posed=fr["joints_posed"], bone_transforms=fr["bone_transforms"], trans=np.array([0.0, 0.0, 3.0], np.float32),
                 root_orient=np.zeros(3, np.float32), pose_body=np.zeros(63, np.float32), pose_hand=np.zeros(6, np.float32))
        v = fr["smpl_verts"] + np.array([0, 0, 0.2], np.float32)
        px = np.round(v[:, :2] / v[:, 2:3] * 300.0 + 128).astype(int)
        sil = np.zeros((H, W), np.uint8)
        ok = (px[:, 0] >= 3) & (px[:, 0] < W - 3) & (px[:, 1] >= 3) & (px[:, 1] < H - 3)
        for dx in range(-3, 4):
            for dy in range(-3, 4):
                sil[px[ok, 1] + dy, px[ok, 0] + dx] = 255
        Image.fromarray(rng.randint(0, 255, (H, W, 3)).astype(np.uint8)).save(sub / "1" / ("%06d.jpg" % f))
        Image.fromarray(sil).save(sub / "1" / ("%06d.png" % f))

    def fake_samples(v, f, w, cmin, cmax, cen, reg, inside, *a, **k):
        gen = torch.Generator(device=v.device).manual_seed(0)
        out = {"points_uniform": torch.rand(1024, 3, device=v.device, generator=gen) * 2 - 1,
               "points_skinning": v[:1024].clone(), "sampled_weights": w[:1024].clone()}
        if inside:
            out["points_inside"] = (torch.rand(1024, 3, device=v.device, generator=gen) - 0.5) * 0.2
        return out

    monkeypatch.setattr(data, "training_samples", fake_samples)
    cfg = config.builtin_config("zju313")
    cfg["model"]["train_smpl"] = True
    cfg["training"].update(out_dir=str(tmp_path / "out"), max_epochs=3, checkpoint_every_n_epochs=1, batch_size=1)
    cfg["data"] = {"dataset": "zju_mocap", "path": str(tmp_path / "data"), "train_split": ["CoreView_000"], "train_views": [],
                   "train_subsampling_rate": 1, "train_start_frame": 0, "train_end_frame": 0, "high_res": False,
                   "num_fg_samples": 256, "num_bg_samples": 128, "off_surface_thr": 0.2, "inside_thr": 0.001, "box_margin": 0.05,
                   "sampling": "default", "sample_reg_surface": True, "erode_mask": True}
    # MetaAvatar-style initialisation files (metaavatar_render/config.py:18-84): the synthetic subject's geometry and
    # skinning networks, so that the first step renders a body instead of an all-zero SDF
    sd = config.synthetic_state_dict(cfg)
    torch.save({"model": {"module.decoder." + k[len("sdf_decoder."):]: v for k, v in sd.items() if k.startswith("sdf_decoder.")}},
               tmp_path / "geo.pt")
    pre = "skinning_model.skinning_decoder_fwd."
    torch.save({"model": {"skinning_decoder_fwd." + k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}}, tmp_path / "skin.pt")
    cfg["model"].update(geometry_net=str(tmp_path / "geo.pt"), skinning_net2=str(tmp_path / "skin.pt"))
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    argv = [str(tmp_path / "cfg.yaml"), "--default-config", str(tmp_path / "cfg.yaml"), "--epochs-per-run", "1"]
    lines = []
    steps = train.main(argv, body=body, faces=np.zeros((1, 3), np.int32), log=lines.append)
    ck = torch.load(tmp_path / "out" / "checkpoints" / "last.ckpt", map_location="cpu")
    assert steps == 2 and ck["epoch"] == 1 and ck["global_step"] == 2
    assert "model.body_poses.pose_body_0" in ck["state_dict"] and "model.betas" in ck["state_dict"]     # train_smpl parameters
    before = ck["state_dict"]["model.color_decoder.lin0.bias"].clone()
    steps = train.main(argv, body=body, faces=np.zeros((1, 3), np.int32), log=lines.append)               # resumes: epoch 1 -> 2
    ck2 = torch.load(tmp_path / "out" / "checkpoints" / "last.ckpt", map_location="cpu")
    assert steps == 4 and ck2["epoch"] == 2 and ck2["global_step"] == 4
    assert not torch.equal(ck2["state_dict"]["model.color_decoder.lin0.bias"], before)                  # it trains
    lm = config.get_model(cfg, mode="test", checkpoint_path=str(tmp_path / "out" / "checkpoints" / "last.ckpt"))
    assert lm.model.latent.num_embeddings == 2
