"""Train a model -- the reference's ``train.py`` (train.py:1-140) without PyTorch Lightning.

    python -m arah_release_amd.train CONFIG.yaml [--epochs-per-run N] [--exit-after SECONDS]
    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 -m arah_release_amd.train CONFIG.yaml

One process per GPU.  What Lightning did for the reference is spelled out here: a rank-sharded shuffled sampler per epoch
(torch DistributedSampler semantics, one view per GPU and step like train.py:46-47), ``LightningModel.training_step`` on
items composed on the device (data.TrainingDataset), one flat all-reduce of the 87 M-parameter gradient on RCCL per step,
Adam with the reference's parameter groups, and checkpoints in Lightning's layout (``epoch``, ``global_step``,
``state_dict`` with the 'model.' prefix, ``optimizer_states``) under ``<out_dir>/checkpoints/last.ckpt`` so that the
reference's test.py / this build's test_sequence load them; ``--epochs-per-run`` chains jobs like train.py:107-122."""
import argparse
import os
import sys
import time

import torch


def build_parser():
    p = argparse.ArgumentParser(description="Training function.")
    p.add_argument("config", type=str, help="Path to config file.")
    p.add_argument("--exit-after", type=int, default=-1, help="Checkpoint and exit after specified number of seconds with exit code 2.")
    p.add_argument("--num-workers", type=int, default=4, help="Accepted for compatibility: items are composed on the GPU.")
    p.add_argument("--epochs-per-run", type=int, default=-1, help="Number of epochs to train before restart.")
    p.add_argument("--run-name", type=str, default="", help="Accepted for compatibility (no wandb in this build).")
    p.add_argument("--default-config", type=str, default="configs/default.yaml")
    p.add_argument("--body-models", type=str, default="body_models/misc", help="Directory of the SMPL model files.")
    return p


def epochs_to_run(max_epochs, epochs_per_run, checkpoint_epoch):
    """train.py:107-122: without --epochs-per-run train to max_epochs; with it, to (epochs already trained) + N."""
    if epochs_per_run <= 0:
        return max_epochs
    if checkpoint_epoch is None:
        return epochs_per_run
    return min(checkpoint_epoch + epochs_per_run, max_epochs)


def epoch_indices(n_items, epoch, rank, world, seed=0):
    """torch.utils.data.DistributedSampler(shuffle=True): a permutation seeded with seed + epoch, padded by wrapping to a
    multiple of the world size, rank r takes positions r, r + world, ..."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n_items, generator=g).tolist()
    total = (n_items + world - 1) // world * world
    idx += idx[:total - n_items]
    return idx[rank:total:world]


def allreduce_gradients(params, world, dist):
    """One flat all-reduce (mean) of every gradient: 348 MB fp32 for the ARAH model, a single RCCL collective per step."""
    grads = [p.grad for p in params if p.grad is not None]
    if world == 1 or not grads:
        return
    flat = torch._utils._flatten_dense_tensors(grads)
    dist.all_reduce(flat)
    flat.div_(world)
    for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(f)


def save_checkpoint(path, lm, opt, epoch, global_step):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = path + ".tmp"
    torch.save({"epoch": epoch, "global_step": global_step,
                "state_dict": {"model." + k: v for k, v in lm.model.state_dict().items()},
                "optimizer_states": [opt.state_dict()]}, tmp)
    os.replace(tmp, path)


def main(argv=None, body=None, faces=None, log=print):
    from . import config, data, smpl
    args = build_parser().parse_args(argv)
    cfg = config.load_config(args.config, args.default_config)
    t_cfg = cfg["training"]
    out_dir = t_cfg["out_dir"]
    if t_cfg.get("batch_size", 1) != 1:
        raise ValueError("one view per GPU and step (batch_size 1), as every ARAH configuration trains")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("training needs a GPU (the renderer has no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    d = cfg["data"]
    if d["dataset"] != "zju_mocap":
        raise ValueError('Invalid dataset "%s" (this build trains on the capture format zju_mocap)' % d["dataset"])
    body = body if body is not None else smpl.BodyModel.from_files("neutral", args.body_models)
    dataset = data.TrainingDataset(
        d["path"], subjects=d["train_split"], mode="train", img_size=(1024, 1024) if d.get("high_res") else (512, 512),
        num_fg_samples=d["num_fg_samples"], num_bg_samples=d["num_bg_samples"], sampling_rate=d["train_subsampling_rate"],
        start_frame=d["train_start_frame"], end_frame=d["train_end_frame"], views=d["train_views"],
        off_surface_thr=d["off_surface_thr"], inside_thr=d["inside_thr"], box_margin=d["box_margin"], sampling=d["sampling"],
        sample_reg_surface=d["sample_reg_surface"], sample_inside=t_cfg.get("inside_weight", 0) > 0, erode_mask=d["erode_mask"],
        body=body, faces=faces, body_models=args.body_models)
    lm = config.get_model(cfg, dataset=dataset, mode="train", body_model=body).to(device)
    lm.train()
    opt = lm.configure_optimizers()
    ckpt_path = os.path.join(out_dir, "checkpoints/last.ckpt")
    epoch0, step, ckpt_epoch = 0, 0, None
    if os.path.exists(ckpt_path):
        ck = torch.load(ckpt_path, map_location="cpu")
        lm.model.load_state_dict({k[6:]: v for k, v in ck["state_dict"].items() if k.startswith("model.")})
        if ck.get("optimizer_states"):
            opt.load_state_dict(ck["optimizer_states"][0])
        epoch0, step, ckpt_epoch = ck["epoch"], ck.get("global_step", 0), ck["epoch"]
    max_epochs = epochs_to_run(t_cfg["max_epochs"], args.epochs_per_run, ckpt_epoch)
    every = t_cfg.get("checkpoint_every_n_epochs", 1)
    params = [p for p in lm.model.parameters() if p.requires_grad]
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    t_start = time.time()
    for epoch in range(epoch0, max_epochs):
        for idx in epoch_indices(len(dataset), epoch, rank, world):
            item = dataset.item(idx, device, generator=gen)
            opt.zero_grad(set_to_none=True)
            losses = lm.compute_loss(item)
            losses["loss"].backward()
            allreduce_gradients(params, world, dist)
            opt.step()
            step += 1
            if rank == 0 and step % 10 == 0:          # log_every_n_steps=10 (train.py:125)
                log("epoch %d step %d " % (epoch, step) + " ".join("%s %.5f" % (k, float(v)) for k, v in losses.items()))
            if args.exit_after > 0 and time.time() - t_start > args.exit_after:
                if rank == 0:
                    save_checkpoint(ckpt_path, lm, opt, epoch, step)
                sys.exit(2)
        if rank == 0 and ((epoch + 1) % every == 0 or epoch + 1 == max_epochs):
            save_checkpoint(ckpt_path, lm, opt, epoch + 1, step)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return step


if __name__ == "__main__":
    main()
