cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r2i/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2i/tests.log 2>&1; tail -30 gpurun_out/r2i/tests.log
