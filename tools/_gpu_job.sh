cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -q -x --timeout=150 2>&1 | tail -3
