"""Timeline of the LAST frame in a rocprofv3 rocpd result: every kernel in start order with its start offset, duration
and the idle gap before it; totals per phase.  A frame starts at k_sort_verts.

    python tools/rocpd_timeline.py <results.db> [--all] > profiles/rNN_timeline.txt
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gcol = next((c for c in ("grid_size_x", "grid_x", "grid_size") if c in cols), None)
wcol = next((c for c in ("workgroup_size_x", "workgroup_x", "workgroup_size") if c in cols), None)
if gcol is None or wcol is None:
    sys.stderr.write("kernels columns: %s\n" % cols)
rows = db.execute("select name, start, end, %s, %s from kernels order by start" % (gcol or "0", wcol or "1")).fetchall()


def short(name):
    s = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*", "", s).replace("void ", "")[:40]


starts = [i for i, r in enumerate(rows) if "k_sort_verts" in r[0]]
i0 = starts[-1]
frame = rows[i0:]
t0 = frame[0][1]
busy = sum(r[2] - r[1] for r in frame)
span = frame[-1][2] - t0
print("last frame: %d kernels, span %.3f ms, kernel time %.3f ms, idle %.3f ms" % (len(frame), span / 1e6, busy / 1e6,
                                                                               (span - busy) / 1e6))
agg = {}
prev_end = t0
gaps = {}
for name, s, e, g, wg in frame:
    k = short(name)
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += max(0, s - prev_end) / 1e3
    prev_end = max(prev_end, e)
print("%-42s %6s %10s %12s" % ("kernel", "calls", "busy_us", "gap_before_us"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-42s %6d %10.1f %12.1f" % (k, a[0], a[1], a[2]))
if "--all" in sys.argv:
    prev_end = t0
    print("\n%10s %9s %8s %8s  %s" % ("t_us", "dur_us", "gap_us", "wgs", "kernel"))
    for name, s, e, g, wg in frame:
        print("%10.1f %9.1f %8.1f %8d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, max(0, s - prev_end) / 1e3, g // max(wg, 1),
                                             short(name)))
        prev_end = max(prev_end, e)
