#!/bin/bash
TAG=${1:-r3az}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-train --passes default > $OUT/bench_prof.json 2> $OUT/prof.err
cd $ROOT
DB=$(ls -S $(find $OUT/prof -name "*.db") | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt && python tools/rocpd_timeline.py $DB --all > $OUT/timeline.txt
rm -rf $OUT/prof
head -3 $OUT/timeline.txt
