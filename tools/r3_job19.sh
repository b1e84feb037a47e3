#!/bin/bash
TAG=${1:-r3t}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
true

timeout 600 python tools/train_regions.py --steps 5 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tee $OUT/train_regions.txt

