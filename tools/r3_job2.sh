#!/bin/bash
# round 3, GPU session 2: ring-distance variants of k_canon_wave, phase clocks, SQ counters
TAG=${1:-r3b}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 -k "broyden3 or tracer_against" > $OUT/tests_canon.log 2>&1
echo "canon tests rc=$?"; tail -3 $OUT/tests_canon.log
B=tools/ubench/bin
timeout 900 python tools/abn.py --rounds 2 base=- lo1=$B/libarah_lo1.so hi0=$B/libarah_hi0.so lo1hi0=$B/libarah_lo1hi0.so 2>&1 | tee $OUT/abn.txt
cd /tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVE_CYCLES"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$N -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-train --passes default > $OUT/pmc_$N.log 2>&1
  S=$(find $OUT/pmc_$N -name "*.db" | head -1)
  [ -n "$S" ] && python $ROOT/tools/rocpd_sq.py $S > $OUT/pmc_$N.json
  rm -rf $OUT/pmc_$N
done
cd $ROOT
python - <<PY
import json
for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS"):
    try:
        d = json.load(open("$OUT/pmc_%s.json" % n))["kernels"]
        for k in ("k_canon_wave<true>", "k_density<true>"):
            print(k, json.dumps(d.get(k)))
    except Exception as e:
        print("pmc parse failed", n, e)
PY
