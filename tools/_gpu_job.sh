cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mesh_query.py tests/test_hip_parity.py -m gpu -q -x -k "mesh or training_samples or reproducible or exports or struct" 2>&1 | tail -15
