"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace --stats) as a per-kernel table.

    python tools/rocpd_stats.py gpurun_out/prof_r1/r1_results.db > profiles/r01_kernel_stats.txt
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                  "max(vgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
print("%-58s %7s %12s %12s %10s %10s %6s %5s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%", "vgpr", "lds"))
for name, n, tot, avg, mn, mx, vg, lds in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short).replace("void ", "")
    print("%-58s %7d %12.3f %12.1f %10.1f %10.1f %6.2f %5s %7s" % (short[:58], n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                  100.0 * tot / total, vg, lds))
print("total GPU kernel time: %.3f ms" % (total / 1e6))
