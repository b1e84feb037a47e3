cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r2l/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2l/tests.log 2>&1; tail -12 gpurun_out/r2l/tests.log
timeout 300 python tools/train_bench.py --steps 5 --warmup 2 > gpurun_out/r2l/train_hip.json 2> gpurun_out/r2l/train_hip.err; cat gpurun_out/r2l/train_hip.json
ARAH_TRAIN_AUTOGRAD=1 timeout 300 python tools/train_bench.py --steps 5 --warmup 2 > gpurun_out/r2l/train_autograd.json 2> gpurun_out/r2l/train_autograd.err; cat gpurun_out/r2l/train_autograd.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2l/prof -o tr -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 3 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r2l/train.err
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/r2l/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB > gpurun_out/r2l/train_stats.txt; head -32 gpurun_out/r2l/train_stats.txt; tail -1 gpurun_out/r2l/train_stats.txt; rm -rf gpurun_out/r2l/prof
