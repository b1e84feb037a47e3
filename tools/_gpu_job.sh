cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2h; mkdir -p $OUT
timeout 120 python tools/train_bench.py --steps 8 --warmup 2 2>/dev/null | cut -c1-220
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 5 --warmup 2 > $OUT/train.json 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/train_kernel_stats.txt
head -12 $OUT/train_kernel_stats.txt | cut -c1-130; tail -1 $OUT/train_kernel_stats.txt
rm -rf $OUT/prof
