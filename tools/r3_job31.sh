#!/bin/bash
TAG=${1:-r3ai}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B=tools/ubench/bin
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=300 -k "nearest or trace or body_tables or forward_against" > $OUT/tests_q.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests_q.log
timeout 900 python tools/abn.py --rounds 3 base=- before=$B/libarah_nn_before.so 2>&1 | tee $OUT/abn.txt
