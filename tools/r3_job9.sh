#!/bin/bash
# round 3, GPU session 9: pipelined halves in the density pass; loop-B finisher default; lattice pin; wide-range subject
TAG=${1:-r3i}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B=tools/ubench/bin
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_meshing.py -m gpu -x -q --timeout=240 -k "wide_range or lattice or lazy_shading or forward_against or reproducible or full_size" > $OUT/tests_q.log 2>&1
echo "quick tests rc=$?"; tail -12 $OUT/tests_q.log
timeout 900 python tools/abn.py --rounds 2 base=- nopp=$B/libarah_nopp.so dens64=-,ARAH_DENSITY_TILE=64 2>&1 | tee $OUT/abn.txt
timeout 900 python -m pytest tests -m gpu -x -q --timeout=240 > $OUT/tests.log 2>&1
echo "pytest rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
