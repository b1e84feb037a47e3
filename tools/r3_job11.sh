#!/bin/bash
TAG=${1:-r3k}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 bash tools/ablate_trunk.sh 2>&1 | grep -v Warning | tee $OUT/trunk_ablation.txt
