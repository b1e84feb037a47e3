cd $GRAFT_REPO_ROOT
for args in "256,192,256,128,256 2 prenosync 0,3,7,11,5" "256,192,256,128,256 3 prenosync 0,3,7,11,5"; do
  timeout 70 python tools/_seq_debug.py $args 2>&1 | grep -v amdgpu.ids | tail -4; echo "--- rc=$? ($args)"
done
timeout 200 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "render_sequence" --timeout=80 --timeout-method=thread 2>&1 | grep -v amdgpu.ids | tail -3
timeout 120 python tools/_pipe_probe.py 3 2>&1 | grep -v amdgpu.ids
