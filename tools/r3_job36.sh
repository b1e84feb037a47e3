#!/bin/bash
TAG=${1:-r3ao}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -k "train or step or F8 or f8 or shade_samples" > $OUT/tests_train.log 2>&1
echo "train tests rc=$?"; tail -2 $OUT/tests_train.log
for h in 1 0; do
ARAH_TRAIN_HANDOVER=$h timeout 300 python tools/train_bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200 | tee $OUT/train_bench_h$h.json
done
