#!/bin/bash
# round 3, GPU session 5: 128-point tiles for the density pass
TAG=${1:-r3e}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 -k "lazy_shading or forward_against or full_size or reproducible" > $OUT/tests_q.log 2>&1
echo "quick tests rc=$?"; tail -3 $OUT/tests_q.log
timeout 900 python tools/abn.py --rounds 2 base=- dens64=-,ARAH_DENSITY_TILE=64 2>&1 | tee $OUT/abn.txt
cd /tmp
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_sq -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-train --passes default > $OUT/pmc_sq.log 2>&1
S=$(find $OUT/pmc_sq -name "*.db" | head -1)
[ -n "$S" ] && python $ROOT/tools/rocpd_sq.py $S > $OUT/pmc_sq.json
rm -rf $OUT/pmc_sq
cd $ROOT
python - <<PY
import json
try:
    d = json.load(open("$OUT/pmc_sq.json"))["kernels"]
    for k in d:
        if k.startswith("k_density") or k.startswith("k_canon"):
            print(k, {a: (round(b, 3) if isinstance(b, float) and b < 100 else b) for a, b in d[k].items() if "frac" in a or "per_" in a or a == "avg_duration_us"})
except Exception as e:
    print("pmc parse failed", e)
PY
