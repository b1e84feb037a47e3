#!/bin/bash
TAG=${1:-r3aj}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1200 python tools/abn.py --rounds 2 base=- bulk8=-,ARAH_TRACE_BULK_STEPS=8 bulk12=-,ARAH_TRACE_BULK_STEPS=12 bulk16=-,ARAH_TRACE_BULK_STEPS=16 bulk24=-,ARAH_TRACE_BULK_STEPS=24 2>&1 | tee $OUT/abn.txt
