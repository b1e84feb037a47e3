"""Is the training step slower after inference frames have run in the same process (bench.py's order)?

profiles/r05c_bench_default.json carries training.ms_per_step = 60.8 where r05a / r05b (before the captured hypernetwork)
carried 25.5 / 25.9.  This probe times bench.GpuRuntime.training_line()
  (a) first thing in a fresh process,
  (b) after a few inference frames with frames in flight (captured hypernetwork, several scratches),
  (c) once more,
and prints torch's allocator figures next to each.  ARAH_HYPERNET_GRAPH=0 gives the eager hypernetwork for an A/B.
"""
import json, os, sys, time
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def mem():
    return {"allocated_GB": round(torch.cuda.memory_allocated() / 2**30, 2), "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 2)}


def main():
    import __graft_entry__
    __graft_entry__.build()
    from arah_release_amd import config, hip, renderer, synthetic
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model, cfg = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
    rt = bench.GpuRuntime(1, 0, dev, None, model, cfg, synthetic.SyntheticScene(0), hip)
    out = {"inference_graph": os.environ.get("ARAH_HYPERNET_GRAPH", "1"), "training_graph": os.environ.get("ARAH_TRAIN_HYPERNET_GRAPH", "0")}
    out["a_fresh"] = dict(rt.training_line(steps=8, warmup=3), **mem())
    frames = [rt.make_inputs(512, k) for k in range(6)]
    with torch.no_grad():
        rt.render_many(frames[:2], 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rt.render_many(frames[2:], 4)
        torch.cuda.synchronize()
        out["inference_ms_per_frame"] = 1e3 * (time.perf_counter() - t0) / 4
    out["b_after_inference"] = dict(rt.training_line(steps=8, warmup=3), **mem())
    out["c_again"] = dict(rt.training_line(steps=8, warmup=3), **mem())
    for k in ("a_fresh", "b_after_inference", "c_again"):
        out[k].pop("note", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
