"""Training step (bench.GpuRuntime.training_line's loop) with the host's share made visible: wall ms per step next to the CPU
time of the process per step.  ARAH_TRAIN_HYPERNET_GRAPH=0|1 picks the eager or the captured hypernetwork; run it under
`taskset -c N` beside a busy loop on the same core to see a host that is half as fast.

    python tools/probes/train_host.py [steps]
"""
import json, os, sys, time
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    import __graft_entry__
    __graft_entry__.build()
    from arah_release_amd import config, synthetic, training
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model, cfg = config.build_synthetic_model("zju313", device=dev)
    model.train()
    opt = training.configure_optimizers(model, cfg)
    crit = training.build_loss(cfg)
    scene = synthetic.SyntheticScene(0)
    warm = 4
    batches = [scene.make_inputs(512, 512, frame_idx=k, max_rays=2048, eval_mode=False, device=dev) for k in range(steps + warm)]

    def step(inp):
        opt.zero_grad(set_to_none=True)
        losses = training.training_step(model, crit, inp)
        losses["loss"].backward()
        opt.step()

    for k in range(warm):
        step(batches[k])
    torch.cuda.synchronize()
    t0, c0 = time.perf_counter(), time.process_time()
    for k in range(warm, warm + steps):
        step(batches[k])
    c1 = time.process_time()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(json.dumps({"graph": os.environ.get("ARAH_TRAIN_HYPERNET_GRAPH", "0"), "ms_per_step": 1e3 * (t1 - t0) / steps,
                      "host_cpu_ms_per_step_until_last_enqueue": 1e3 * (c1 - c0) / steps, "steps": steps}))
    if os.environ.get("TRAIN_HOST_CPROFILE"):   # where the Python side of the step goes (the autograd engine's thread is not seen)
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        for k in range(warm, warm + steps):
            step(batches[k])
        pr.disable()
        torch.cuda.synchronize()
        for key in ("tottime", "cumulative"):
            print("==== cProfile of %d steps, sorted by %s" % (steps, key))
            pstats.Stats(pr).strip_dirs().sort_stats(key).print_stats(70)


if __name__ == "__main__":
    main()
