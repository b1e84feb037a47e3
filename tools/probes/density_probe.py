"""The density pass (loop D, pass 1) of a few benchmark frames: SHA-256 of the rendered outputs and the event-timed duration
of the pass.  Run once per build / knob (ARAH_DENSITY_REG=0|1 picks the tile kernel or the point-owning trunk) and compare
the lines: same hashes = same bits.

    python tools/probes/density_probe.py [n_frames] [size]
"""
import hashlib, json, os, sys, time
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    import __graft_entry__
    __graft_entry__.build()
    from arah_release_amd import config, hip, synthetic
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model, cfg = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
    rt = bench.GpuRuntime(1, 0, dev, None, model, cfg, synthetic.SyntheticScene(0), hip)
    rt.set_adaptive(False)
    frames = [rt.make_inputs(size, k) for k in range(n_frames)]
    out = {"density_reg": os.environ.get("ARAH_DENSITY_REG", "0"), "frames": []}
    with torch.no_grad():
        rt.render(dict(frames[0]))
        torch.cuda.synchronize()
        rt.reset_counters()
        rt.set_events(False, True)
        for f in frames:
            o = rt.render(dict(f))
            torch.cuda.synchronize()
            h = hashlib.sha256()
            for k in ("rgb_values", "network_body_mask", "points_cam"):
                h.update(o[k].detach().cpu().numpy().tobytes())
            out["frames"].append({"sha256": h.hexdigest()[:16], "density_ms": rt.event_ms(), "canon_ms": rt.canon_ms(),
                                  "rays": int(f["ray_dirs"].shape[1])})
        rt.set_events(False, False)
        c = rt.counters()
        out["n_density"] = c["n_density"]
        out["n_col"] = c["n_col"]
        if os.environ.get("RT_CLOCKS"):   # instrumented build (-DRT_CLOCKS through ARAH_LIB_PATH): s_memtime ticks per wave and phase
            import ctypes as C
            lib = hip.load_library()
            ws = rt.tracer.workspace(dev)
            clk = (C.c_ulonglong * 256)()
            rc = lib.arah_debug_clocks(C.c_void_p(ws.buf.data_ptr()), clk, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc
            names = ["load+L0", "k1", "k2", "k3", "k4", "k5", "tail+head", "store"]
            out["clocks_pct"] = {}
            for w in range(4):
                tot = sum(clk[w * 16 + i] for i in range(16))
                out["clocks_pct"]["wave%d" % w] = {n: round(100.0 * clk[w * 16 + i] / max(tot, 1), 1) for i, n in enumerate(names)}
                out["clocks_pct"]["wave%d" % w]["Gticks"] = round(tot / 1e9, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
