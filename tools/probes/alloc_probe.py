"""Does the timed inference region of bench.py make device allocations / frees (torch's caching allocator going to the driver)?
Counts them over consecutive eight-frame passes with frames in flight."""
import json, os, sys, time
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import __graft_entry__
    __graft_entry__.build()
    from arah_release_amd import config, hip, synthetic
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model, cfg = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
    rt = bench.GpuRuntime(1, 0, dev, None, model, cfg, synthetic.SyntheticScene(0), hip)
    frames = [rt.make_inputs(512, k) for k in range(10)]
    out = []
    with torch.no_grad():
        rt.render_many(frames[:2], 4)
        torch.cuda.synchronize()
        for rep in range(4):
            s0 = torch.cuda.memory_stats(dev)
            t0, c0 = time.perf_counter(), time.process_time()
            rt.render_many(frames[2:], 4)
            c1 = time.process_time()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            s1 = torch.cuda.memory_stats(dev)
            out.append({"ms_per_frame": round(1e3 * dt / 8, 2), "host_cpu_ms_per_frame": round(1e3 * (c1 - c0) / 8, 2),
                        "device_alloc": int(s1.get("num_device_alloc", 0) - s0.get("num_device_alloc", 0)),
                        "device_free": int(s1.get("num_device_free", 0) - s0.get("num_device_free", 0)),
                        "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 2)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
