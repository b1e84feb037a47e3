"""tools/mfma_adjacent.py <isa.s | disassembly> [kernel-substring]: MFMAs whose A / B operand is written by a VALU instruction
fewer wait states ahead than the MI355X needs (measured, tools/ubench/valu_mfma_hazard.hip and valu_mfma32_hazard.hip,
profiles/r04_hazard_ubench.txt):
    v_mfma_f32_16x16x32_f16 (and the other 16-bit MFMAs)  1 wait state behind a VALU write of its B register,
    v_mfma_f32_16x16x4_f32                                 2 (an s_waitcnt with nothing to wait for counts as one).
hipcc inserts them for instructions it can see.  It cannot see into an inline-asm statement: round 3's T blend selected its B
operand with asm v_cndmask_b32 and, in schedules that put the MFMA right behind it, points skinned to joints 12..15 missed their
roots.  `violations(text)` is what tests/test_hip_parity.py::test_device_code_keeps_mfma_operand_distance runs on the shipped
code object."""
import re
import sys


def _regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"[va](\d+)$", tok)
    if m:
        return {int(m.group(1))}
    return set()


def _instr(line):
    """instruction text of an assembly (.s) or llvm-objdump -d line, or None"""
    s = line.split("//")[0].strip()
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":") or s.startswith("<"):
        return None
    if re.match(r"^[0-9a-f]+ <", s):
        return None
    return s


def violations(text, pat=""):
    out = []
    name, window = None, []   # window: (wait states this instruction provides, instruction)
    for n, ln in enumerate(text.splitlines(), 1):
        m = re.match(r"^(_Z\w+):", ln) or re.match(r"^[0-9a-f]+ <(_Z\w+)>:", ln)
        if m:
            name, window = m.group(1), []
            continue
        s = _instr(ln)
        if s is None or name is None or pat not in name:
            continue
        op = s.split()[0]
        if op.startswith("v_mfma"):
            need = 2 if re.search(r"x\d+_f32$|x\d+f32$", op) else 1
            ops = re.split(r",\s*", s[len(op):].strip())
            src = set()
            for i in (1, 2):
                if i < len(ops):
                    src |= _regs(ops[i].split()[0])
            dist = 0
            for ws, prev in reversed(window):
                if dist >= need:
                    break
                pop = prev.split()[0]
                if pop.startswith("v_") and not pop.startswith("v_mfma") and not pop.startswith("v_cmp") and not pop.startswith("v_readfirstlane"):
                    ptoks = re.split(r",\s*", prev[len(pop):].strip())
                    if _regs(ptoks[0]) & src:
                        out.append((name, n, dist, prev, s))
                        break
                dist += ws
        ws = 1
        m = re.match(r"s_nop\s+(\d+)", s)
        if m:
            ws = int(m.group(1)) + 1
        window.append((ws, s))
        if len(window) > 8:
            window.pop(0)
    return out


if __name__ == "__main__":
    v = violations(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "")
    for name, n, dist, a, b in v[:40]:
        print("%s line %d: %d wait state(s) between\n      %s\n      %s" % (name[:90], n, dist, a, b))
    print("%d MFMA operand(s) written too close ahead" % len(v))
