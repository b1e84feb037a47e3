cd $GRAFT_REPO_ROOT
timeout 250 python -m pytest tests/test_callers.py -m gpu -q -x --timeout=120 2>&1 | grep -v amdgpu.ids | tail -25
