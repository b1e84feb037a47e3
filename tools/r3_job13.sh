#!/bin/bash
# round 3, GPU session 13: sixteen-lane nearest search for the ray lists; pipelined trunk GEMM (variants)
TAG=${1:-r3n}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B=tools/ubench/bin
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 -k "nearest or trace or lazy_shading or forward_against or full_size or reproducible" > $OUT/tests_q.log 2>&1
echo "quick tests rc=$?"; tail -3 $OUT/tests_q.log
for v in pipe2 pipe4; do
  ARAH_LIB_PATH=$B/libarah_$v.so timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 -k "forward_against or full_size" > $OUT/tests_$v.log 2>&1
  echo "$v tests rc=$?"; tail -1 $OUT/tests_$v.log
done
timeout 1200 python tools/abn.py --rounds 2 base=- nogroup=-,ARAH_KNN_GROUP=0 group_all=-,ARAH_KNN_WAVE_RAYS=100000000 group64k=-,ARAH_KNN_WAVE_RAYS=65536 pipe2=$B/libarah_pipe2.so pipe4=$B/libarah_pipe4.so 2>&1 | tee $OUT/abn.txt
