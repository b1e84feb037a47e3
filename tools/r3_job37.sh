#!/bin/bash
TAG=${1:-r3aq}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for r in 1 2 3; do for h in 1 0; do
echo -n "handover=$h "; ARAH_TRAIN_HANDOVER=$h timeout 300 python tools/train_bench.py --steps 20 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['peak_mem_gb'],2))"
done; done | tee $OUT/ab.txt
