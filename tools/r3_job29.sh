#!/bin/bash
TAG=${1:-r3af}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=300 > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?"; tail -3 $OUT/tests_gpu.log
timeout 300 python tools/train_bench.py --steps 8 --warmup 2 2>&1 | tail -1 | cut -c1-200 | tee $OUT/train_bench.json
ARAH_TRACE_SMALL=0 timeout 300 python tools/train_bench.py --steps 8 --warmup 2 2>&1 | tail -1 | cut -c1-200 | tee $OUT/train_bench_bulk.json
