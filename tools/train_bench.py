"""Training-step benchmark (BASELINE.json configs[2]: ZJUMOCAP-313 shapes, 1 view x 2048 rays per GPU, data-parallel
gradient all-reduce on RCCL).

    python tools/train_bench.py --steps 5                       # one GPU
    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 tools/train_bench.py --steps 5

One step = forward (HIP ray tracer under no_grad + autograd loop D / regularisers) + IDHRLoss + backward (+ DDP
all-reduce of the 87 M-parameter gradient, 348 MB fp32) + Adam.  Prints one JSON line on rank 0.
"""
import argparse, json, os, sys, time
import torch
import torch.distributed as dist

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="zju313")
    ap.add_argument("--exchange", default="native", choices=["native", "ddp"],
                    help="native: arah_release_amd.train.GradientExchange (the product's train entry: fixed bucket order, "
                         "presence map for parameters a rank did not use); ddp: torch DistributedDataParallel with "
                         "find_unused_parameters=True (what the reference's Lightning strategy='ddp' amounts to)")
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from arah_release_amd import config, synthetic, training, train
    torch.manual_seed(0)
    model, cfg = config.build_synthetic_model(args.config, device=dev)
    model.train()
    exchange = None
    if world > 1 and args.exchange == "ddp":
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], find_unused_parameters=True)
    else:
        net = model
        train.broadcast_state(model, world, dist)
        exchange = train.GradientExchange([p for p in model.parameters() if p.requires_grad], world, dist)
    opt = training.configure_optimizers(model, cfg)
    crit = training.build_loss(cfg)
    scene = synthetic.SyntheticScene(0)
    # every rank sees a different view (here: frame) per step, 2048 rays each (configs/default.yaml:13-14)
    batches = [scene.make_inputs(512, 512, frame_idx=rank + world * k, max_rays=2048, eval_mode=False, device=dev)
               for k in range(args.steps + args.warmup)]

    def step(inp):
        opt.zero_grad(set_to_none=True)
        losses = training.training_step(net, crit, inp)
        losses["loss"].backward()
        if exchange is not None:
            exchange.finish()
        opt.step()
        return losses

    for k in range(args.warmup):
        step(batches[k])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        losses = step(batches[k])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "training rays/sec", "value": 2048 * world * args.steps / float(t), "unit": "rays/s",
                          "n_gpus": world, "steps": args.steps, "ms_per_step": 1e3 * float(t) / args.steps,
                          "config": {"workload": "%s training step, 1 view x 2048 rays per GPU, synthetic" % args.config},
                          "loss": float(losses["loss"]), "peak_mem_gb": torch.cuda.max_memory_allocated() / 2**30}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
