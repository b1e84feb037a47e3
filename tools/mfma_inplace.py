"""tools/mfma_inplace.py <isa.s> [kernel-substring]: lists the MFMAs of a kernel whose SrcC is a register range other than
their destination.  Round 4 met wrong roots in loop C that came and went with the schedule; every failing build had such
MFMAs with the SrcC registers rewritten a few instructions later (the compiler inserts the 3 wait states it believes a 4-pass
XDL needs), every build without them was correct (profiles/r04_srcc_war.txt)."""
import re
import sys

src = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
name = None
bad = {}
tot = {}
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name = m.group(1)
        continue
    if name is None or pat not in name:
        continue
    m = re.match(r"\s+v_mfma_\w+ (\S+), (\S+), (\S+), (\S+)", ln)
    if m:
        tot[name] = tot.get(name, 0) + 1
        d, c = m.group(1).rstrip(","), m.group(4).split()[0]
        if c != "0" and c != d:
            bad[name] = bad.get(name, 0) + 1
for k in tot:
    print("%-110s mfma %4d  out-of-place SrcC %4d" % (k[:110], tot[k], bad.get(k, 0)))
