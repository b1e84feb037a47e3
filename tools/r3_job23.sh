#!/bin/bash
TAG=${1:-r3z}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for e in b3 fp32; do
ARAH_TRAIN_ENGINE=$e timeout 900 python -m pytest tests -m gpu -q --timeout=300 -k "train or step or shade_samples or F8 or f8" > $OUT/tests_train_$e.log 2>&1
echo "train tests ($e) rc=$?"; tail -6 $OUT/tests_train_$e.log
ARAH_TRAIN_ENGINE=$e timeout 300 python tools/train_bench.py --steps 8 --warmup 2 2>&1 | tail -1 | cut -c1-200 | tee $OUT/train_bench_$e.json
done
