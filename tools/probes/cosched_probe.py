"""Do loop C's solver (vector-ALU bound) and the density pass (matrix-pipe bound) run faster SIDE BY SIDE on the same CUs than
one after the other?  (the 2-GPU-minute A/B that comes before any restructuring.)

One process = one library build + one set of knobs (ARAH_LIB_PATH, ARAH_CANON_LDS_MIN, ARAH_DENSITY_TILE, ARAH_MAX_GRID,
ARAH_CANON_KERNEL); the driver mode runs the arms one after the other on the same box and prints a table.

    python tools/probes/cosched_probe.py            # driver: all arms
    python tools/probes/cosched_probe.py --arm      # one arm with the environment as it is (prints one JSON line)

Per arm, on one 512 x 512 x 64 frame of the benchmark workload:
  canon_alone    arah_sample_canonicalize (sampler, nearest-vertex search, loop C's solver) on stream A
  shade_alone    arah_shade_composite (density pass, k_shade on the sigma > 0 samples, compositing) on stream B, on the
                 canonical samples of a frame rendered before
  seq            both on stream A, one after the other
  par            A and B enqueued at the same time (what frames in flight do when their phases meet)
  par_canon_first  B enqueued when loop C's solver has started on A
and the HIP-event durations of loop C's solver and of the density pass inside each of them.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def arm():
    import torch
    from arah_release_amd import config, hip, synthetic
    dev = torch.device("cuda:0")
    model, cfg = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
    scene = synthetic.SyntheticScene(0)
    tracer = model.idhr_network.ray_tracer
    reps = int(os.environ.get("COSCHED_REPS", "5"))
    with torch.no_grad():
        inp = scene.make_inputs(512, 512, frame_idx=0, device=dev)
        model(inp, eval=True)
        frame = model.idhr_network.last_frame
        B, N, _ = inp["ray_dirs"].shape
        cam = inp["cam_loc"].reshape(B, 3)
        d = inp["ray_dirs"].reshape(B * N, 3).contiguous()
        nf = inp["body_bounds_intersections"].reshape(B * N, 2).contiguous()
        ws_a, ws_b = hip.Workspace(dev), hip.Workspace(dev)
        samp_a = hip.Sampling(dev, 64, 16, 16, model.idhr_network.cano_view_dirs, model.idhr_network.render_last_pt)
        samp_b = hip.Sampling(dev, 64, 16, 16, model.idhr_network.cano_view_dirs, model.idhr_network.render_last_pt)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        samp_a.set_events("canon", ev[0], ev[1])
        samp_b.set_events("density", ev[2], ev[3])
        xn, T, conv, start, end = hip.trace(frame, ws_a, cam, d, nf)
        z, pts, Ts, mask = hip.sample_canonicalize(frame, ws_a, samp_a, cam, d, nf, conv, start, end)
        rgb, acc, vol = hip.shade_composite(frame, ws_b, samp_b, d, z, pts, Ts, mask)
        torch.cuda.synchronize()
        check = {"pts_sum": float(pts[mask.bool()].double().sum()), "mask": int(mask.sum()), "rgb_sum": float(rgb.double().sum())}
        sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

        def canon(st):
            with torch.cuda.stream(st):
                return hip.sample_canonicalize(frame, ws_a, samp_a, cam, d, nf, conv, start, end)

        def shade(st):
            with torch.cuda.stream(st):
                return hip.shade_composite(frame, ws_b, samp_b, d, z, pts, Ts, mask)

        def timed(fn):
            out = []
            for r in range(reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                keep = fn()
                torch.cuda.synchronize()
                dt = 1e3 * (time.perf_counter() - t0)
                if r:   # the first repetition warms the allocator
                    out.append((dt, ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])))
                del keep
            med = sorted(out)[len(out) // 2]
            return {"wall_ms": med[0], "canon_kernel_ms": med[1], "density_kernel_ms": med[2], "all_wall_ms": [round(o[0], 2) for o in out]}

        # the shader clock while the arms run (tools/ubench/clock_trace.hip: one wave logs s_memtime against the 100 MHz
        # s_memrealtime every 0.25 ms on a third stream)
        import ctypes
        ctl = os.path.join(ROOT, "tools", "ubench", "bin", "libclock_trace.so")
        ct = ctypes.CDLL(ctl) if os.path.exists(ctl) else None
        sc = torch.cuda.Stream(dev)
        n_smp, period = 240, 25000
        cbuf = torch.zeros(2 * n_smp, dtype=torch.int64, device=dev)

        def clocks(fn):
            if ct is None:
                return None
            import numpy as np
            torch.cuda.synchronize()
            ct.clock_trace_launch(ctypes.c_void_p(sc.cuda_stream), ctypes.c_void_p(cbuf.data_ptr()), n_smp, period)
            t0 = time.perf_counter()
            keep = fn()
            sa.synchronize()
            sb.synchronize()
            dt = 1e3 * (time.perf_counter() - t0)
            torch.cuda.synchronize()
            del keep
            a = cbuf.cpu().numpy().reshape(n_smp, 2).astype(np.float64)
            t_ms = (a[1:, 0] - a[0, 0]) * 1e-5
            mhz = (a[1:, 1] - a[:-1, 1]) / np.maximum(a[1:, 0] - a[:-1, 0], 1.0) * 100.0
            busy = (t_ms > 2.0) & (t_ms < dt - 1.0)
            return {"wall_ms": dt, "busy_mhz_mean": float(mhz[busy].mean()) if busy.any() else None,
                    "busy_mhz_min": float(mhz[busy].min()) if busy.any() else None,
                    "idle_mhz_mean": float(mhz[t_ms > dt + 3.0].mean()) if (t_ms > dt + 3.0).any() else None,
                    "mhz_every_2ms": [int(x) for x in mhz[::8]]}

        res = {"canon_alone": timed(lambda: canon(sa)),
               "shade_alone": timed(lambda: shade(sb)),
               "seq": timed(lambda: (canon(sa), shade(sa))),
               "par": timed(lambda: (canon(sa), shade(sb))),
               "par_shade_first": timed(lambda: (shade(sb), canon(sa))),
               # the shading call enqueued once loop C's solver has STARTED (the sampler and the nearest-vertex search in
               # front of it -- 119 KB of LDS per workgroup -- cannot share a CU with the density pass: enqueued together, the
               # search waits for the whole density pass and the two calls run one after the other)
               "par_canon_first": timed(lambda: (canon(sa), ev[0].synchronize(), shade(sb)))}
        res["canon_alone"]["clock"] = clocks(lambda: canon(sa))
        res["shade_alone"]["clock"] = clocks(lambda: shade(sb))
        res["par_canon_first"]["clock"] = clocks(lambda: (canon(sa), ev[0].synchronize(), shade(sb)))
        res["canon_alone"].pop("density_kernel_ms")
        res["shade_alone"].pop("canon_kernel_ms")
    knobs = {k: os.environ[k] for k in ("ARAH_LIB_PATH", "ARAH_CANON_LDS_MIN", "ARAH_DENSITY_TILE", "ARAH_MAX_GRID",
                                        "ARAH_CANON_KERNEL", "ARAH_CANON_WG_PER_CU") if k in os.environ}
    print(json.dumps({"knobs": knobs, "n_rays": int(B * N), "check": check, "arms": res}))


ARMS = [
    ("default (8-wave canon, hi halves in LDS; 128-point density)", {}),
    ("8-wave canon all-L2; 64-point density, 2 WG/CU", {"ARAH_CANON_KERNEL": "wave_l2", "ARAH_DENSITY_TILE": "64"}),
    ("HALF canon (4 waves, all-L2, 84 KB => 1/CU); 64-point density capped at 1 WG/CU",
     {"ARAH_LIB_PATH": "tools/ubench/bin/libarah_half.so", "ARAH_CANON_KERNEL": "wave_l2", "ARAH_CANON_LDS_MIN": "86016",
      "ARAH_DENSITY_TILE": "64", "ARAH_MAX_GRID": "256"}),
    ("HALF canon (4 waves, all-L2, 84 KB => 1/CU); 64-point density, grid 512",
     {"ARAH_LIB_PATH": "tools/ubench/bin/libarah_half.so", "ARAH_CANON_KERNEL": "wave_l2", "ARAH_CANON_LDS_MIN": "86016",
      "ARAH_DENSITY_TILE": "64"}),
    ("HALF canon, 2 WG/CU (24 KB each); 64-point density",
     {"ARAH_LIB_PATH": "tools/ubench/bin/libarah_half.so", "ARAH_CANON_KERNEL": "wave_l2", "ARAH_CANON_WG_PER_CU": "2",
      "ARAH_DENSITY_TILE": "64"}),
]


def driver():
    rows = []
    only = os.environ.get("COSCHED_ARMS")
    for i, (name, env) in enumerate(ARMS):
        if only and str(i) not in only.split(","):
            continue
        e = dict(os.environ)
        for k, v in env.items():
            e[k] = os.path.join(ROOT, v) if k == "ARAH_LIB_PATH" else v
        if "ARAH_LIB_PATH" in e and not os.path.exists(e["ARAH_LIB_PATH"]):
            print("skip (library not built):", name)
            continue
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm"], env=e, capture_output=True, text=True)
        try:
            r = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print("FAILED", name, out.stderr[-1500:])
            continue
        rows.append((name, r))
        print("==", name)
        print("   check", r["check"])
        for k, v in r["arms"].items():
            print("   %-16s wall %6.2f ms   loop C %s   density %s   %s" % (
                k, v["wall_ms"], "%6.2f" % v["canon_kernel_ms"] if "canon_kernel_ms" in v else "   -  ",
                "%6.2f" % v["density_kernel_ms"] if "density_kernel_ms" in v else "   -  ", v["all_wall_ms"]))
            if v.get("clock"):
                c = v["clock"]
                print("   %-16s shader clock while busy: mean %s MHz, min %s MHz (idle afterwards %s); every 2 ms: %s" % (
                    "", c["busy_mhz_mean"] and int(c["busy_mhz_mean"]), c["busy_mhz_min"] and int(c["busy_mhz_min"]),
                    c["idle_mhz_mean"] and int(c["idle_mhz_mean"]), c["mhz_every_2ms"]))
        sys.stdout.flush()
    print(json.dumps({n: r for n, r in rows}))


if __name__ == "__main__":
    arm() if "--arm" in sys.argv else driver()
