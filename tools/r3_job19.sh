#!/bin/bash
TAG=${1:-r3t}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q --timeout=300 -k "train or step or shade_samples or F8 or f8" > $OUT/tests_train.log 2>&1
echo "train tests rc=$?"; tail -3 $OUT/tests_train.log
timeout 600 python tools/train_regions.py --steps 5 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tee $OUT/train_regions.txt
timeout 300 python tools/train_bench.py --steps 8 --warmup 2 2>&1 | tail -1 | tee $OUT/train_bench.json
