#!/bin/bash
# round 3, GPU session 16: group search on the LDS table for the sample list; ablation ladder of the pipelined trunk
TAG=${1:-r3q}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B=tools/ubench/bin
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 > $OUT/tests_q.log 2>&1
echo "parity tests rc=$?"; tail -3 $OUT/tests_q.log
timeout 900 python tools/abn.py --rounds 2 base=- thread_samples=-,ARAH_KNN_GROUP_SAMPLES=0 2>&1 | tee $OUT/abn.txt
{
ARAH_DENSITY_TILE=128 python tools/ablate_trunk.py --density
for v in NO_EPI NO_BARRIER A_FIXED B_FIXED AB_FIXED ALL; do
  ARAH_DENSITY_TILE=128 ARAH_LIB_PATH=$B/libarah_abl_$v.so python tools/ablate_trunk.py --density
done
} 2>&1 | grep -v "amdgpu.ids\|Warning" | tee $OUT/trunk_ablation_pipe.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python $ROOT/bench.py --steps 3 --warmup 1 --pipelined-streams 1 --no-cpu-baseline --no-train --passes default > $OUT/prof.log 2>&1
S=$(ls -S $(find $OUT/prof -name "*.db") | head -1)
[ -n "$S" ] && python $ROOT/tools/rocpd_stats.py $S > $OUT/kernel_stats.txt && python $ROOT/tools/rocpd_timeline.py $S --all > $OUT/timeline.txt
rm -rf $OUT/prof
head -30 $OUT/timeline.txt
