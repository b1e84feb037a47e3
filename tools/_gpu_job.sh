cd $GRAFT_REPO_ROOT
python tools/ab.py tools/ubench/bin/libarah_prev.so arah_release_amd/libarah_hip.so 3 2>&1 | grep -v amdgpu.ids
