cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh r2c pmc 2>&1 | grep -v amdgpu.ids | head -24
python tools/phase_clocks.py run 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c/phase_clocks.txt
