#!/bin/bash
TAG=${1:-r3r}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for gs in 0 1; do
ARAH_KNN_GROUP_SAMPLES=$gs timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout=240 -k "tracer or nearest" > $OUT/tests_gs$gs.log 2>&1
echo "GROUP_SAMPLES=$gs rc=$?"; tail -4 $OUT/tests_gs$gs.log
done
