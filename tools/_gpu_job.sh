cd $GRAFT_REPO_ROOT
python tools/ab.py tools/ubench/bin/libarah_prev.so arah_release_amd/libarah_hip.so 3 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "sdf_eval or skin or tracer or forward_against or lazy" 2>&1 | tail -3
