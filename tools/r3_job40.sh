#!/bin/bash
TAG=${1:-r3au}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q --timeout=300 -k "composite_samples or sdf_normal or train or step or F8 or f8 or shade_samples" > $OUT/tests_train.log 2>&1
echo "train tests rc=$?"; tail -12 $OUT/tests_train.log | cut -c1-220
for r in 1 2; do for h in 1 0; do
echo -n "run=$h "; ARAH_UNUSED=$h timeout 300 python tools/train_bench.py --steps 20 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['loss'])"
done; done | tee $OUT/ab.txt
