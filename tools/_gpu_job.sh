cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_callers.py -m gpu -q -x --timeout=100 -k "validation_step or test_sequence" 2>&1 | grep -v amdgpu.ids | tail -20
