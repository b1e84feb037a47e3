cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh r2y pmc 2>&1 | grep -v amdgpu.ids | head -40
python tools/phase_clocks.py run 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r2y/phase_clocks.txt
