#!/bin/bash
TAG=${1:-r3ay}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=300 > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?"; tail -3 $OUT/tests_gpu.log
timeout 300 python tools/train_bench.py --steps 20 --warmup 4 2>/dev/null | tail -1 | cut -c1-200
timeout 600 python tools/abn.py --rounds 2 base=- 2>&1 | tail -2
