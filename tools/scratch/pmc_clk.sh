export TMPDIR=/tmp; mkdir -p gpurun_out/pmc3
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d gpurun_out/pmc3 -o clk -- python tools/perf_seams.py zju377_mono 2e6 > gpurun_out/pmc3/clk.log 2>&1
tail -1 gpurun_out/pmc3/clk.log
