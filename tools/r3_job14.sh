#!/bin/bash
# round 3, GPU session 14: group search as the only ray search; pipelined trunk GEMM at both tile widths
TAG=${1:-r3o}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B=tools/ubench/bin
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 -k "nearest or trace or lazy_shading or forward_against or full_size or reproducible" > $OUT/tests_q.log 2>&1
echo "quick tests rc=$?"; tail -3 $OUT/tests_q.log
for v in pipe2_44 pipe2_42; do
  ARAH_LIB_PATH=$B/libarah_$v.so timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 > $OUT/tests_$v.log 2>&1
  echo "$v tests rc=$?"; tail -1 $OUT/tests_$v.log
done
timeout 1500 python tools/abn.py --rounds 2 base=- pipe2=$B/libarah_pipe2.so p44=$B/libarah_pipe2_44.so p42=$B/libarah_pipe2_42.so p44d64=$B/libarah_pipe2_44.so,ARAH_DENSITY_TILE=64 p42d64=$B/libarah_pipe2_42.so,ARAH_DENSITY_TILE=64 2>&1 | tee $OUT/abn.txt
