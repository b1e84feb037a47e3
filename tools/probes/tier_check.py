"""Tiered against untiered eval forward on benchmark frames: bit identity of rgb / mask / acc / points_cam, violations
({rays with some density > 0 in the exact path} must be a subset of {surface or promoted rays}), tier statistics, time per frame.

    python tools/probes/tier_check.py [--frames 0,1,5,8] [--size 512] [--n-steps 64] [--config zju377_mono] [--time]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="0,1,2,3,5,8,13,17")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--n-steps", type=int, default=64)
    ap.add_argument("--config", default="zju377_mono")
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--beta", type=float, default=None)
    args = ap.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    from arah_release_amd import config, synthetic
    dev = torch.device("cuda", 0)
    near = far = args.n_steps // 4
    model, cfg = config.build_synthetic_model(args.config, args.n_steps, near, far, device=dev)
    if args.beta is not None:
        with torch.no_grad():
            model.deviation_decoder.variance.fill_(args.beta)
    idhr = model.idhr_network
    idhr.adaptive_shading = False
    scene = synthetic.SyntheticScene(0)
    tracer = idhr.ray_tracer
    rows = []
    for fi in [int(x) for x in args.frames.split(",")]:
        inputs = scene.make_inputs(args.size, args.size, frame_idx=fi, device=dev)
        n = inputs["ray_dirs"].shape[1]
        res = {}
        for mode in ("exact", "tiered"):
            idhr.tiering = mode == "tiered"
            with torch.no_grad():
                ws = tracer.workspace(dev)
                if ws.buf is not None:
                    ws.reset_counters()
                out = model(dict(inputs), eval=True)
                torch.cuda.synchronize()
                ws = tracer.workspace(dev)
                tier, pos = ws.tier_debug(n, args.n_steps)
                ctr = ws.counters()
                t_ms = None
                if args.time:
                    for _ in range(2):
                        model(dict(inputs), eval=True)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        model(dict(inputs), eval=True)
                    torch.cuda.synchronize()
                    t_ms = (time.perf_counter() - t0) / 5 * 1e3
            res[mode] = {"rgb": out["rgb_values"].clone(), "mask": out["network_body_mask"].clone(), "pcam": out["points_cam"].clone(),
                         "tier": tier.clone(), "pos": pos.clone(), "ctr": ctr, "ms": t_ms,
                         "info": ws.occupancy_info() if mode == "tiered" else None}
        e, t = res["exact"], res["tiered"]
        same_rgb = bool(torch.equal(e["rgb"], t["rgb"]))
        same_mask = bool(torch.equal(e["mask"], t["mask"]))
        same_pcam = bool(torch.equal(e["pcam"], t["pcam"]))
        n_diff = int((e["rgb"] != t["rgb"]).any(-1).sum())
        n_mask_diff = int((e["mask"] != t["mask"]).sum())
        viol = int(((e["pos"] == 1) & (t["tier"] == 0)).sum())
        c = t["ctr"]
        row = {"frame": fi, "rays": n, "same_rgb": same_rgb, "same_mask": same_mask, "same_points_cam": same_pcam, "rays_rgb_differ": n_diff,
               "rays_mask_differ": n_mask_diff, "violations": viol, "exact_rays_with_sigma": int((e["pos"] == 1).sum()),
               "tier": {k: v for k, v in c.items() if k.startswith("n_tier") or k.endswith("_p2")},
               "n_canon": (e["ctr"]["n_canon"], c["n_canon"]), "n_density": (e["ctr"]["n_density"], c["n_density"]),
               "n_knn": (e["ctr"]["n_knn"], c["n_knn"]), "n_col": (e["ctr"]["n_col"], c["n_col"]),
               "ms": (e["ms"], t["ms"]), "occ": t["info"]}
        rows.append(row)
        print(json.dumps(row), flush=True)
    ok = all(r["same_rgb"] and r["same_mask"] and r["same_points_cam"] and r["violations"] == 0 for r in rows)
    print("ALL IDENTICAL, ZERO VIOLATIONS" if ok else "MISMATCH", flush=True)


if __name__ == "__main__":
    main()
