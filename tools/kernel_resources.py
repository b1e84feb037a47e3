"""Print VGPR / scratch / LDS / occupancy per kernel of csrc/arah_hip.hip (hipcc -Rpass-analysis)."""
import os
import re
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "arah_release_amd", "csrc", "arah_hip.hip")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-fno-slp-vectorize", "-c", src, "-o", "/tmp/arah_res.o",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    if "error" in line:
        print(line)
    m2 = re.search(r"Function Name: (\S+)", line)
    if m2:
        cur = {"name": m2.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
print("%-46s %6s %6s %8s %6s %9s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r in rows:
    name = r["name"]
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
    except Exception:
        pass
    print("%-46s %6s %6s %8s %6s %9s" % (name[:46], r.get("VGPRs", "?"), r.get("AGPRs", "?"),
                                           r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"),
                                           r.get("LDS Size [bytes/block]", "?")))
