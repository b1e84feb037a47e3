"""tools/mfma_adjacent.py <isa.s> [kernel-substring]: MFMAs whose A / B / C operand is written by the VALU instruction right
in front of them.  tools/ubench/valu_mfma_hazard.hip: on gfx950 a v_mfma_f32_16x16x32_f16 issued in the cycle after a VALU
write of its B operand reads the OLD register (distance 0 wrong, one wait state right); this lists such pairs."""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"[va](\d+)$", tok)
    if m:
        return {int(m.group(1))}
    return set()


src = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
name, prev, prev_ln = None, None, 0
found = {}
for n, ln in enumerate(src, 1):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name, prev = m.group(1), None
        continue
    s = ln.strip()
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
        continue
    if name is None or pat not in name:
        continue
    op = s.split()[0]
    if op.startswith("v_mfma") and prev is not None:
        pop = prev.split()[0]
        if pop.startswith("v_") and not pop.startswith("v_mfma") and not pop.startswith("v_cmp"):
            ops = [t for t in re.split(r",\s*", s[len(op):].strip())]
            ptoks = re.split(r",\s*", prev[len(pop):].strip())
            wr = regs(ptoks[0])
            for i, what in ((1, "A"), (2, "B"), (3, "C")):
                if i < len(ops) and wr & regs(ops[i].split()[0]):
                    found.setdefault(name, []).append((prev_ln, what, prev, s))
    prev, prev_ln = s, n
for k, v in found.items():
    print(k[:100], len(v), "pairs")
    for ln, what, a, b in v[:12]:
        print("   line %d  src%s:  %s   ->   %s" % (ln, what, a, b))
if not found:
    print("no VALU write directly in front of an MFMA that reads it")
