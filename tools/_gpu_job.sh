cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
python tools/_tol_probe.py 2>&1 | grep -v Warning | tee gpurun_out/r2j/tol.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2j/prof -o tr -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r2j/train.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2j/train.err
cd $GRAFT_REPO_ROOT
cat gpurun_out/r2j/train.json
DB=$(find gpurun_out/r2j/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB > gpurun_out/r2j/train_stats.txt; head -40 gpurun_out/r2j/train_stats.txt; tail -1 gpurun_out/r2j/train_stats.txt; rm -rf gpurun_out/r2j/prof
