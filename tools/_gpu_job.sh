cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r2k/build.log 2>&1
timeout 600 python -m pytest tests -m gpu -q -x -k "shade_samples_op or struct_sizes" > gpurun_out/r2k/tests.log 2>&1; tail -40 gpurun_out/r2k/tests.log
