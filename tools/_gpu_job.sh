cd $GRAFT_REPO_ROOT
timeout 300 python tools/train_bench.py --steps 8 --warmup 2 2>/dev/null | cut -c1-200
timeout 600 python -m pytest tests -m gpu -q -x -k "training or shade_samples or gram" 2>&1 | tail -3
timeout 600 python tools/_prof_train.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 > gpurun_out/prof_train.txt; head -45 gpurun_out/prof_train.txt | cut -c1-45,120-260
