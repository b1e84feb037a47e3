"""Where one training step goes:  python tools/train_regions.py [--steps 5]
Times the regions of tools/train_bench.py's step with a device synchronisation on both sides of each (so the regions
do not overlap: their sum is larger than the free-running step) and counts the GPU kernels each one launches
(torch.profiler, one step).  Prints a table; profiles/r03_train_regions.txt is its output on the MI355X."""
import argparse, collections, os, sys, time
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

TIMES = collections.OrderedDict()
STACK = []


PROF = [False]


def region(name):
    if PROF[0]:
        return torch.profiler.record_function("R:" + name)

    class R:
        def __enter__(self):
            torch.cuda.synchronize()
            self.t0 = time.perf_counter()
            STACK.append([name, 0.0])

        def __exit__(self, *a):
            torch.cuda.synchronize()
            dt = time.perf_counter() - self.t0
            _, inner = STACK.pop()
            if STACK:
                STACK[-1][1] += dt
            TIMES.setdefault(name, [0.0, 0.0, 0])
            TIMES[name][0] += dt            # inclusive
            TIMES[name][1] += dt - inner    # exclusive of nested regions
            TIMES[name][2] += 1
    return R()


def wrap(obj, attr, name):
    f = getattr(obj, attr)

    def g(*a, **k):
        with region(name):
            return f(*a, **k)
    setattr(obj, attr, g)


def main():
    os.environ.setdefault("ARAH_TRAIN_HYPERNET_GRAPH", "0")   # the regions wrap the decoder with synchronising calls: no capture
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="zju313")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    import __graft_entry__
    __graft_entry__.build()
    from arah_release_amd import config, synthetic, training, renderer, hip
    torch.manual_seed(0)
    model, cfg = config.build_synthetic_model(args.config, device=dev)
    model.train()
    opt = training.configure_optimizers(model, cfg)
    crit = training.build_loss(cfg)
    scene = synthetic.SyntheticScene(0)
    batches = [scene.make_inputs(512, 512, frame_idx=k, max_rays=2048, eval_mode=False, device=dev)
               for k in range(args.steps + args.warmup)]
    idhr = model.idhr_network
    wrap(model.sdf_decoder, "forward", "fwd: pose encoder + hypernetwork (sdf_decoder)")
    wrap(idhr, "forward_train", "fwd: renderer.forward_train (all of the below)")
    wrap(renderer, "build_frame", "fwd:   build_frame (pack weights, body prep)")
    wrap(idhr.ray_tracer, "forward", "fwd:   ray tracer, loops A-C (HIP, no_grad)")
    wrap(training, "query_weights", "fwd:   skinning-weight query (points_skinning)")
    wrap(training, "shade_composite_train", "fwd:   loop D + compositing (shade_composite_train)")
    wrap(training, "shade_samples_hip", "fwd:     ShadeSamples forward (HIP)")
    wrap(training, "forward_skinning", "fwd:     forward_skinning re-attachment")
    wrap(hip, "skin_jacobian", "fwd:     skin_jacobian (HIP)")

    def step(inp, timed):
        if not timed:
            opt.zero_grad(set_to_none=True)
            training.training_step(model, crit, inp)["loss"].backward()
            opt.step()
            return
        with region("step"):
            with region("zero_grad"):
                opt.zero_grad(set_to_none=True)
            with region("forward (model)"):
                out = model(inp)
            with region("loss"):
                gt = {"rgb": inp["rgb_values"]}
                if "sampled_weights" in inp:
                    gt["sampled_weights"] = inp["sampled_weights"]
                losses = crit(out, gt)
            with region("backward"):
                losses["loss"].backward()
            with region("optimizer"):
                opt.step()

    import copy
    for k in range(args.warmup):
        step(batches[k], False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        step(copy.copy(batches[k]), False)
    torch.cuda.synchronize()
    free = (time.perf_counter() - t0) / args.steps
    TIMES.clear()
    for k in range(args.warmup, args.warmup + args.steps):
        step(copy.copy(batches[k]), True)
    print("free-running step: %.2f ms" % (1e3 * free))
    print("%-62s %10s %10s %6s" % ("region (synchronised on both sides)", "incl ms", "excl ms", "calls"))
    for name, (inc, exc, n) in TIMES.items():
        print("%-62s %10.2f %10.2f %6.1f" % (name, 1e3 * inc / args.steps, 1e3 * exc / args.steps, n / args.steps))
    # kernel launches per region, one step under the profiler (regions become record_function ranges, no syncs)
    from torch.profiler import profile, ProfilerActivity
    PROF[0] = True
    inp = copy.copy(batches[args.warmup])
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(inp, True)
        torch.cuda.synchronize()
    ev = prof.events()
    ranges = [(e.name[2:], e.time_range.start, e.time_range.end) for e in ev if e.name.startswith("R:")]
    launches = [e for e in ev if "LaunchKernel" in e.name or e.name in ("hipMemcpyAsync", "hipMemsetAsync", "hipMemcpyWithStream")]
    syncs = [e for e in ev if e.name in ("hipStreamSynchronize", "hipDeviceSynchronize", "hipMemcpyWithStream", "hipEventSynchronize")]
    counts, cpu_us, nsync = collections.Counter(), collections.Counter(), collections.Counter()
    for name, a, b in ranges:
        cpu_us[name] += b - a
    for coll, dst in ((launches, counts), (syncs, nsync)):
        for e in coll:
            best = None
            for name, a, b in ranges:   # innermost enclosing range
                if a <= e.time_range.start <= b and (best is None or b - a < best[1]):
                    best = (name, b - a)
            dst[best[0] if best else "(outside)"] += 1
    print("%-62s %10s %10s %8s" % ("region (innermost; one free-running step under torch.profiler)", "launches", "host syncs", "cpu ms"))
    for name in TIMES:
        print("%-62s %10d %10d %8.2f" % (name, counts[name], nsync[name], cpu_us[name] / 1e3))
    print("total launches", len(launches), " host syncs", len(syncs))
    # backward: launches by autograd node
    nodes = [(e.name.split(": ", 1)[1], e.time_range.start, e.time_range.end) for e in ev
             if e.name.startswith("autograd::engine::evaluate_function: ")]
    nodes.sort(key=lambda t: t[1])
    import bisect
    starts = [t[1] for t in nodes]
    by_node = collections.Counter()
    n_calls = collections.Counter(t[0] for t in nodes)
    for e in launches:
        k = bisect.bisect_right(starts, e.time_range.start) - 1
        if k >= 0 and nodes[k][1] <= e.time_range.start <= nodes[k][2]:
            by_node[nodes[k][0]] += 1
    print("backward launches by autograd node (node: launches / node calls):")
    for name, c in by_node.most_common(40):
        print("  %-50s %6d / %d" % (name[:50], c, n_calls[name]))
    # forward: launches by aten op (top level ops only approximated by innermost aten:: range)
    ops = [(e.name, e.time_range.start, e.time_range.end) for e in ev if e.name.startswith("aten::")]
    fwd = [r for r in ranges if r[0].startswith("forward (model)")]
    if fwd:
        a0, b0 = fwd[0][1], fwd[0][2]
        top = collections.Counter()
        ops_f = sorted([o for o in ops if a0 <= o[1] <= b0], key=lambda t: t[1])
        # outermost aten ops
        outer, end = [], -1
        for o in ops_f:
            if o[1] > end:
                outer.append(o)
                end = o[2]
        st = [o[1] for o in outer]
        for e in launches:
            if a0 <= e.time_range.start <= b0:
                k = bisect.bisect_right(st, e.time_range.start) - 1
                if k >= 0 and outer[k][1] <= e.time_range.start <= outer[k][2]:
                    top[outer[k][0]] += 1
                else:
                    top["(custom / HIP seam)"] += 1
        print("forward launches by outermost aten op:")
        for name, c in top.most_common(25):
            print("  %-50s %6d" % (name[:50], c))


if __name__ == "__main__":
    main()
