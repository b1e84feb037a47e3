#!/bin/bash
# round 3, GPU session 8: finishers (local counters; standard / fast trunk), wide-range skinning subject, lattice pin
TAG=${1:-r3h}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_meshing.py -m gpu -x -q --timeout=240 -k "wide_range or lattice or broyden3 or joint_root or tracer_against or forward_against" > $OUT/tests_q.log 2>&1
echo "quick tests rc=$?"; tail -12 $OUT/tests_q.log
timeout 900 python tools/abn.py --rounds 2 base=- old=-,ARAH_TRACE_BULK_STEPS=50,ARAH_JOINT_BULK_ITERS=51 fast=-,ARAH_FINISH_FAST=1 onlyB=-,ARAH_TRACE_BULK_STEPS=50 2>&1 | tee $OUT/abn.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-train --passes default > $OUT/bench_prof.json 2> $OUT/prof.err
cd $ROOT
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt && head -16 $OUT/kernel_stats.txt
rm -rf $OUT/prof
