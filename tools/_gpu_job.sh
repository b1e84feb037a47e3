cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x --timeout=120 2>&1 | tail -3
python tools/ab.py tools/ubench/bin/libarah_prev.so arah_release_amd/libarah_hip.so 2 2>&1 | grep -v amdgpu.ids
