#!/bin/bash
# round 3, GPU session 10: hypernetwork output layers through arah_gemv_rows; full tests; profile of the default pass
TAG=${1:-r3j}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 -k "gemv or forward_against or sdf_eval" > $OUT/tests_q.log 2>&1
echo "quick tests rc=$?"; tail -6 $OUT/tests_q.log
timeout 900 python tools/abn.py --rounds 2 base=- nogemv=-,ARAH_HYPER_GEMV=0 2>&1 | tee $OUT/abn.txt
timeout 900 python -m pytest tests -m gpu -x -q --timeout=240 > $OUT/tests.log 2>&1
echo "pytest rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --streams 1 --pipelined-streams 1 --no-cpu-baseline --no-train --passes default > $OUT/bench_prof.json 2> $OUT/prof.err
cd $ROOT
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt && head -24 $OUT/kernel_stats.txt
[ -n "$DB" ] && python tools/rocpd_timeline.py $DB --all > $OUT/timeline.txt && head -3 $OUT/timeline.txt
rm -rf $OUT/prof
