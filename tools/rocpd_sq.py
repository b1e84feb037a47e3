"""Per-kernel SQ counter averages from one rocprofv3 --pmc pass (rocpd SQLite output).

    python tools/rocpd_sq.py gpurun_out/<tag>/pmc_SQ.../..._results.db > profiles/rNN_pmc_sq.json

Derived figures (units per MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
waves, SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles):
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * 4 SIMDs / n_SE-normalisation) is NOT attempted here --
  the raw sums are reported next to the ratios that need no normalisation:
    valu_active_per_wave_cycle = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES
    wait_inst_per_wave_cycle   = SQ_WAIT_INST_ANY   / SQ_WAVE_CYCLES
    mfma_busy_per_sq_busy      = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES
"""
import json
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
acc = {}
for name, counter, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                        "group by kernel_name, counter_name"):
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short).replace("void ", "")
    if not short.startswith("k_"):
        continue
    acc.setdefault(short, {"launches": n})[counter] = avg
# launch durations from the kernel trace of the same pass -> MFMA pipe occupancy
dur = {}
try:
    for name, avg in db.execute("select name, avg(duration) from kernels group by name"):
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short).replace("void ", "")
        dur[short] = avg
except sqlite3.OperationalError:
    pass
for k, v in acc.items():
    if k in dur:
        v["avg_duration_us"] = dur[k] / 1e3
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:   # cycles summed over the 1024 SIMDs / (duration x clock x 1024)
            v["mfma_busy_frac_at_2.4GHz"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur[k] * 2.4 * 1024)
            v["mfma_busy_frac_at_2.0GHz"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur[k] * 2.0 * 1024)
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        for c, out in (("SQ_ACTIVE_INST_VALU", "valu_active_per_wave_cycle"), ("SQ_WAIT_INST_ANY", "wait_inst_per_wave_cycle"),
                       ("SQ_ACTIVE_INST_ANY", "active_inst_per_wave_cycle")):
            if c in v:
                v[out] = v[c] / wc
    if v.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        v["mfma_busy_per_sq_busy"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CYCLES"]
print(json.dumps({"units": "SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over all SIMDs (16 per v_mfma_f32_16x16x32_f16, 32 per "
                           "v_mfma_f32_16x16x4_f32 x4 K-steps...); mfma_busy_frac = that / (launch duration x clock x 1024 SIMDs), "
                           "given at the 2.4 GHz peak clock and at 2.0 GHz (the chip runs 1.9-2.3 GHz under MFMA load, "
                           "MI355X_MICROARCH.md)",
                  "note": "per-launch averages of one rocprofv3 --pmc pass over bench.py --steps 1 --warmup 1 --passes default",
                  "kernels": acc}, indent=1))
