cd $GRAFT_REPO_ROOT
timeout 150 python -m pytest tests/test_callers.py -m gpu -q -x --timeout=100 2>&1 | grep -v amdgpu.ids | tail -25
