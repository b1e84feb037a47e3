cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_train_entry.py -m gpu -q -x --timeout=150 2>&1 | grep -v amdgpu.ids | tail -30
