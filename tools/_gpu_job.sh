cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2o/tests.log 2>&1; tail -15 gpurun_out/r2o/tests.log
timeout 600 python bench.py > gpurun_out/r2o/bench.json 2> gpurun_out/r2o/bench.err; tail -3 gpurun_out/r2o/bench.err; cat gpurun_out/r2o/bench.json
