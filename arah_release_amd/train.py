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

import numpy as np
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


class GradientExchange:
    """The data-parallel gradient exchange of one training step (what DistributedDataParallel with
    find_unused_parameters=True did for the reference's `strategy='ddp'`, train.py:131).

    * The parameter list and its partition into buckets (~25 MB, in reverse registration order: the order gradients
      become ready in) are the same on every rank, whatever subset of the parameters a rank's item touched -- per-frame
      SMPL parameters (train_smpl, models/__init__.py:117-123) get a gradient only on the rank that drew their frame.
    * A bucket whose gradients have all arrived is all-reduced at once, from the gradient hooks, while backward() is
      still running: the 348 MB exchange overlaps the backward pass.  Collectives are matched by issue order, so buckets
      go out strictly in bucket order on every rank: a bucket holding a parameter that did not fire on this rank, and
      every bucket after it, goes out in finish(), the missing gradients zero-filled.
    * finish() also sums a presence map: a parameter that received a gradient on NO rank keeps grad None (Adam skips it,
      like under DDP); every other parameter gets the mean over ALL ranks (absent contributions count as zero).  The
      last entry of the map carries the ranks' stop requests (--exit-after), so that all ranks leave together."""

    def __init__(self, params, world, dist, bucket_bytes=25 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.world, self.dist = world, dist
        self.order = list(range(len(self.params)))[::-1]
        self.buckets, cur, size = [], [], 0
        for i in self.order:
            n = self.params[i].numel() * self.params[i].element_size()
            if cur and size + n > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(i)
            size += n
        if cur:
            self.buckets.append(cur)
        self.bucket_of, self.offset = {}, {}
        self.flat = []
        for b, idx in enumerate(self.buckets):
            off = 0
            for i in idx:
                self.bucket_of[i], self.offset[i] = b, off
                off += self.params[i].numel()
            p0 = self.params[idx[0]]
            self.flat.append(torch.zeros(off, dtype=p0.dtype, device=p0.device))
        self.hooks = []
        # On the GPU (RCCL) the collectives are issued from ONE stream of this object's own, ordered by events behind the copies
        # that fill a bucket: a gradient hook runs on whatever stream autograd replays that node on, and the exchange must not
        # depend on which.  (ProcessGroupNCCL makes its internal stream wait for the stream that is current at the call.)
        self.comm_stream = torch.cuda.Stream(self.params[0].device) if self.params and self.params[0].is_cuda else None
        if world > 1:
            for i, p in enumerate(self.params):
                self.hooks.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._ready(i)))
        self.begin()

    def begin(self):
        self.fired = [False] * len(self.params)
        self.missing = [len(idx) for idx in self.buckets]
        self.work = [None] * len(self.buckets)
        self.copied = [False] * len(self.buckets)
        self.filled = [[] for _ in self.buckets]   # GPU: events behind the copies into each bucket
        self.next = 0        # first bucket that has not been sent yet

    def _slice(self, i):
        p = self.params[i]
        return self.flat[self.bucket_of[i]][self.offset[i]:self.offset[i] + p.numel()]

    def _ready(self, i):
        if self.fired[i]:
            return          # (a second accumulation into the same parameter would need a second exchange: not the case here)
        self.fired[i] = True
        b = self.bucket_of[i]
        self.missing[b] -= 1
        if self.missing[b] == 0:
            self._fill(b)
        while self.next < len(self.buckets) and self.missing[self.next] == 0:
            self._send(self.next)
            self.next += 1

    def _fill(self, b):
        """The gradients of bucket b's parameters that have arrived -> its flat buffer, the others zero: ONE multi-tensor copy
        (and one fill) per bucket.  Round 6: a copy and an event per PARAMETER (211 of each, and 211 copies back in finish())
        were ~450 launches on top of a training step that is bound by its ~660."""
        if self.copied[b]:
            return
        self.copied[b] = True
        have = [i for i in self.buckets[b] if self.fired[i] and self.params[i].grad is not None]
        miss = [i for i in self.buckets[b] if i not in set(have)]
        if have:
            torch._foreach_copy_([self._slice(i) for i in have], [self.params[i].grad.reshape(-1) for i in have])
        if miss:
            torch._foreach_zero_([self._slice(i) for i in miss])
        if self.comm_stream is not None:
            self.filled[b].append(torch.cuda.current_stream(self.params[0].device).record_event())

    def _send(self, b):
        """All-reduce bucket b (asynchronously), behind everything that wrote into it."""
        if self.comm_stream is None:
            self.work[b] = self.dist.all_reduce(self.flat[b], async_op=True)
            return
        for ev in self.filled[b]:
            self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            self.work[b] = self.dist.all_reduce(self.flat[b], async_op=True)

    def finish(self, stop=False):
        """Call after backward().  Returns True if any rank asked to stop."""
        if self.world == 1:
            self.begin()
            return bool(stop)
        dev = self.params[0].device
        for i, p in enumerate(self.params):                       # gradients set without a hook firing (tests, manual .grad)
            if not self.fired[i] and p.grad is not None:
                self._ready(i)
        for b in range(self.next, len(self.buckets)):
            self._fill(b)                                         # what arrived, the rest zero
            self._send(b)
        self.next = len(self.buckets)
        seen = torch.tensor([1.0 if f else 0.0 for f in self.fired] + [1.0 if stop else 0.0], device=dev)
        seen_work = self.dist.all_reduce(seen, async_op=True)
        seen_work.wait()
        seen = seen.tolist()
        for b in range(len(self.buckets)):
            self.work[b].wait()
        torch._foreach_div_(self.flat, float(self.world))
        dst, src = [], []
        for i, p in enumerate(self.params):
            if seen[i] > 0:
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                dst.append(p.grad)
                src.append(self._slice(i).view_as(p))
        if dst:
            torch._foreach_copy_(dst, src)                        # the means back into the .grad tensors, one multi-tensor copy
        self.begin()
        return bool(seen[-1] > 0)


def allreduce_gradients(params, world, dist):
    """One-shot form of GradientExchange for gradients that are already in place (no overlap with backward)."""
    if world == 1:
        return
    ex = GradientExchange(params, world, dist)
    for h in ex.hooks:
        h.remove()
    ex.finish()


def broadcast_state(module, world, dist):
    """Rank 0's parameters and buffers to every rank (what Lightning's DDP does when it wraps the model): the parts that
    are not loaded from files -- the colour network, the latent codes, most of the hypernetwork -- are drawn from each
    process's own generator."""
    if world == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, 0)
    from .renderer import invalidate_caches
    invalidate_caches(module)   # the broadcast wrote through .data: version counters did not move


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
    # im2mesh/config.py:141-250: the three capture formats and their image-size rules (training mode)
    high_res = bool(d.get("high_res"))
    kinds = {"zju_mocap": (data.TrainingDataset, (1024, 1024) if high_res else (512, 512)),
             "h36m": (data.H36MDataset, (1002, 1000)),
             "people_snapshot": (data.PeopleSnapshotDataset, (1080, 1080) if high_res else (540, 540))}
    if d["dataset"] not in kinds:
        raise ValueError('Invalid dataset "%s"' % d["dataset"])
    cls, img_size = kinds[d["dataset"]]
    dataset = cls(
        d["path"], subjects=d["train_split"], mode="train", img_size=img_size,
        num_fg_samples=d["num_fg_samples"], num_bg_samples=d["num_bg_samples"], sampling_rate=d["train_subsampling_rate"],
        start_frame=d["train_start_frame"], end_frame=d["train_end_frame"], views=d["train_views"],
        off_surface_thr=d["off_surface_thr"], inside_thr=d["inside_thr"], box_margin=d["box_margin"], sampling=d["sampling"],
        sample_reg_surface=d["sample_reg_surface"], sample_inside=t_cfg.get("inside_weight", 0) > 0, erode_mask=d["erode_mask"],
        body=body, faces=faces, body_models=args.body_models)
    body = dataset.body   # the subject's gendered body model (People-Snapshot), the neutral one otherwise
    torch.manual_seed(0)
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
    broadcast_state(lm.model, world, dist)
    # From here on the global generators drive per-step draws only (eikonal probe points, stratified jitter, pose / view input
    # noise): every rank and every resumed run gets its own stream, like the unseeded ranks of the reference.  (The common seed
    # above only made construction repeatable; broadcast_state has made it redundant.)
    # (+ 7919 * step: a run stopped mid-epoch resumes in the same epoch at a later step and must not replay the epoch's draws)
    torch.manual_seed(4321 + rank + 1000 * (ckpt_epoch or 0) + 7919 * step)
    np.random.seed((4321 + rank + 1000 * (ckpt_epoch or 0) + 7919 * step) % (2 ** 32))
    max_epochs = epochs_to_run(t_cfg["max_epochs"], args.epochs_per_run, ckpt_epoch)
    every = t_cfg.get("checkpoint_every_n_epochs", 1)
    params = [p for p in lm.model.parameters() if p.requires_grad]
    exchange = GradientExchange(params, world, dist)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank + 1000 * (ckpt_epoch or 0) + 7919 * step)   # pixel sampling: a resumed run (also one stopped mid-epoch: same epoch, later step) draws a new sequence
    t_start = time.time()
    for epoch in range(epoch0, max_epochs):
        for idx in epoch_indices(len(dataset), epoch, rank, world):
            item = dataset.item(idx, device, generator=gen)
            opt.zero_grad(set_to_none=True)
            losses = lm.compute_loss(item)
            losses["loss"].backward()           # buckets of gradients are all-reduced from the hooks while this runs
            stop = exchange.finish(stop=args.exit_after > 0 and time.time() - t_start > args.exit_after)
            opt.step()
            step += 1
            if rank == 0 and step % 10 == 0:          # log_every_n_steps=10 (train.py:125)
                log("epoch %d step %d " % (epoch, step) + " ".join("%s %.5f" % (k, float(v)) for k, v in losses.items()))
            if stop:                                  # a collective decision: every rank sees the same flag at the same step
                if rank == 0:
                    save_checkpoint(ckpt_path, lm, opt, epoch, step)
                if world > 1:
                    dist.barrier()
                    dist.destroy_process_group()
                sys.exit(2)
        if rank == 0 and ((epoch + 1) % every == 0 or epoch + 1 == max_epochs):
            save_checkpoint(ckpt_path, lm, opt, epoch + 1, step)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return step


if __name__ == "__main__":
    main()
