#!/bin/bash
# the shade-everything path's HBM traffic: FETCH_SIZE / WRITE_SIZE passes with ARAH_FULL_SHADING=1
TAG=${1:-r3bb}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  ARAH_FULL_SHADING=1 timeout 120 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmcfull_$C -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-train --passes default > $OUT/pmcfull_$C.log 2>&1
done
cd $ROOT
big() { ls -S $(find $1 -name "*.db") 2>/dev/null | head -1; }
F=$(big $OUT/pmcfull_FETCH_SIZE); W=$(big $OUT/pmcfull_WRITE_SIZE)
[ -n "$F" ] && [ -n "$W" ] && python tools/rocpd_pmc.py $F $W > $OUT/pmc_traffic_full.json
rm -rf $OUT/pmcfull_FETCH_SIZE $OUT/pmcfull_WRITE_SIZE
python - <<PY
import json
try:
    k = json.load(open("$OUT/pmc_traffic_full.json"))["kernels"]
    for n, v in k.items():
        if n.startswith("k_shade") or n.startswith("k_density"): print(n, v["launches"], v["hbm_bytes_avg"])
except Exception as e:
    print("no summary", e)
PY
