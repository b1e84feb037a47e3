cd $GRAFT_REPO_ROOT
timeout 60 tools/ubench/bin/hw_sin_accuracy | tee gpurun_out/hw_sin_accuracy.txt
timeout 600 python -m pytest tests -m gpu -q -x --timeout=120 2>&1 | tail -4
python tools/ab.py tools/ubench/bin/libarah_prev.so arah_release_amd/libarah_hip.so 2 2>&1 | grep -v amdgpu.ids
