cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests/test_meshing.py -m gpu -q -x > gpurun_out/r2n/tests.log 2>&1; tail -30 gpurun_out/r2n/tests.log
