#!/bin/bash
# round 3, GPU session 15: pipelined GEMM default; two-half pipeline (pp) on the explicit schedule
TAG=${1:-r3p}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B=tools/ubench/bin
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 -k "nearest or trace or lazy_shading or forward_against or full_size or reproducible" > $OUT/tests_q.log 2>&1
echo "quick tests rc=$?"; tail -3 $OUT/tests_q.log
ARAH_LIB_PATH=$B/libarah_pp.so timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 > $OUT/tests_pp.log 2>&1
echo "pp tests rc=$?"; tail -3 $OUT/tests_pp.log
timeout 1500 python tools/abn.py --rounds 3 base=- pp=$B/libarah_pp.so rolled=$B/libarah_rolled.so 2>&1 | tee $OUT/abn.txt
