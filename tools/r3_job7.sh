#!/bin/bash
# round 3, GPU session 7: finishers with the latency-optimised 16-point trunk; new parity fixtures on the device
TAG=${1:-r3g}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_meshing.py -m gpu -x -q --timeout=240 -k "joint_root or tracer_against or forward_against or full_size or edge_cases or lattice" > $OUT/tests_q.log 2>&1
echo "quick tests rc=$?"; tail -5 $OUT/tests_q.log
timeout 900 python tools/abn.py --rounds 2 base=- old=-,ARAH_TRACE_BULK_STEPS=50,ARAH_JOINT_BULK_ITERS=51 a10b3=-,ARAH_TRACE_BULK_STEPS=10,ARAH_JOINT_BULK_ITERS=3 a14b2=-,ARAH_TRACE_BULK_STEPS=14,ARAH_JOINT_BULK_ITERS=2 2>&1 | tee $OUT/abn.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-train --passes default > $OUT/bench_prof.json 2> $OUT/prof.err
cd $ROOT
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt && head -16 $OUT/kernel_stats.txt
[ -n "$DB" ] && python tools/rocpd_timeline.py $DB --all > $OUT/timeline.txt && head -3 $OUT/timeline.txt
rm -rf $OUT/prof
