cd $GRAFT_REPO_ROOT
SKIP_TESTS=1 bash tools/gpu_round.sh r2v pmc 2>&1 | grep -v amdgpu.ids | head -20
python tools/phase_clocks.py run 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r2v/phase_clocks.txt
