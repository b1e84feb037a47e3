cd $GRAFT_REPO_ROOT
timeout 600 python tools/_dead_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
