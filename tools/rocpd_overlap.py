"""How busy is the GPU with several frames in flight?  From a rocprofv3 rocpd result of a multi-stream run: over the steady
part of the trace (between the first and the last k_canon_wave), the wall time, the time at least one kernel is running, the
mean number of kernels running, and per kernel its total time and the share of it spent alone on the GPU.

    python tools/rocpd_overlap.py <results.db>
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()


def short(name):
    s = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*", "", s).replace("void ", "")[:44]


cw = [r for r in rows if "k_canon_wave<true, false" in r[0] or "k_canon_wave<(bool)1, (bool)0" in r[0]]
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else None
t_lo = cw[len(cw) // 3][1]          # skip the warm-up third
t_hi = cw[-1][2]
ev = []
for name, s, e in rows:
    if e <= t_lo or s >= t_hi:
        continue
    ev.append((max(s, t_lo), 1, name))
    ev.append((min(e, t_hi), -1, name))
ev.sort(key=lambda x: (x[0], x[1]))
active = {}
depth = 0
busy = 0
area = 0
alone = {}
total = {}
prev = t_lo
for t, d, name in ev:
    dt = t - prev
    if depth > 0:
        busy += dt
        area += dt * depth
        for k, c in active.items():
            if c > 0:
                total[k] = total.get(k, 0) + dt * c
                if depth == c:
                    alone[k] = alone.get(k, 0) + dt
    prev = t
    k = short(name)
    active[k] = active.get(k, 0) + d
    depth += d
wall = t_hi - t_lo
frames = sum(1 for r in cw if r[1] >= t_lo) / 2.0   # two solver launches per tiered frame
print("window %.1f ms, ~%.1f frames (%.2f ms per frame); >= 1 kernel running %.1f %% of it; mean kernels running while busy %.2f"
      % (wall / 1e6, frames, wall / 1e6 / max(frames, 1), 100.0 * busy / wall, area / max(busy, 1)))
print("%-46s %10s %10s %8s" % ("kernel", "ms/frame", "alone ms/f", "alone %"))
for k, v in sorted(total.items(), key=lambda kv: -kv[1])[:28]:
    print("%-46s %10.3f %10.3f %7.0f%%" % (k, v / 1e6 / frames, alone.get(k, 0) / 1e6 / frames, 100.0 * alone.get(k, 0) / v))
