"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), rocpd SQLite output.

    python tools/rocpd_pmc.py gpurun_out/pmc/fetch_results.db gpurun_out/pmc/write_results.db > profiles/r01_pmc_traffic.json

Units and corrections as MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB
(bytes = value * 1024); on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so the read
side is doubled ("fetch_bytes_corrected"); WRITE_SIZE is uncalibrated and used as is.
"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    out = {}
    for name, n, avg, mx in db.execute("select kernel_name, count(*), avg(value), max(value) from counters_collection "
                                       "where counter_name=? group by kernel_name", (counter,)):
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short).replace("void ", "")
        out[short] = {"launches": n, "avg_bytes": avg * 1024.0, "max_bytes": mx * 1024.0}
    return out


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f, w = fetch.get(k), write.get(k)
    res[k] = {"launches": (f or w)["launches"],
              "fetch_bytes_raw_avg": f and f["avg_bytes"], "fetch_bytes_corrected_avg": f and 2.0 * f["avg_bytes"],
              "write_bytes_avg": w and w["avg_bytes"],
              "hbm_bytes_avg": (2.0 * f["avg_bytes"] if f else 0.0) + (w["avg_bytes"] if w else 0.0)}
print(json.dumps({"note": "bench.py --steps 1 --warmup 1 (lazy pass then full-shading pass, 3 frames each); "
                          "bytes per launch, FETCH_SIZE doubled per the gfx950 correction", "kernels": res}, indent=1))
