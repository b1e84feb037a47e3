cd $GRAFT_REPO_ROOT
python tools/ab.py arah_release_amd/libarah_hip.so tools/ubench/bin/libarah_nofence.so 3 2>&1 | grep -v amdgpu.ids
