export TMPDIR=/tmp; mkdir -p gpurun_out/pmc2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU -d gpurun_out/pmc2 -o sq -- python tools/perf_seams.py zju377_mono 2e6 > gpurun_out/pmc2/sq.log 2>&1
tail -3 gpurun_out/pmc2/sq.log
