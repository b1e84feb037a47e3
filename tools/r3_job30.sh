#!/bin/bash
TAG=${1:-r3ag}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=300 -k "shade_composite or forward_against or lazy or full_size or reproducible or split_engine" > $OUT/tests_q.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests_q.log
timeout 900 python tools/abn.py --rounds 2 base=- shade_fp32=-,ARAH_SHADE_ENGINE=fp32 2>&1 | tee $OUT/abn.txt
