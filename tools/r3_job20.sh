#!/bin/bash
TAG=${1:-r3w}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 300 python tools/train_bench.py --steps 8 --warmup 2 2>&1 | tail -1 | tee $OUT/train_bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python $ROOT/tools/train_bench.py --steps 5 --warmup 2 > $OUT/prof.log 2>&1
S=$(ls -S $(find $OUT/prof -name "*.db") | head -1)
[ -n "$S" ] && python $ROOT/tools/rocpd_stats.py $S > $OUT/train_kernel_stats.txt
rm -rf $OUT/prof
head -45 $OUT/train_kernel_stats.txt
