#!/bin/bash
TAG=${1:-r3ak}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
ABN_STREAMS=3 timeout 1200 python tools/abn.py --rounds 2 base=- bulk24=-,ARAH_TRACE_BULK_STEPS=24 bulk32=-,ARAH_TRACE_BULK_STEPS=32 2>&1 | tee $OUT/abn3.txt
timeout 1200 python tools/abn.py --rounds 2 base=- bulk20=-,ARAH_TRACE_BULK_STEPS=20 bulk32=-,ARAH_TRACE_BULK_STEPS=32 2>&1 | tee $OUT/abn1.txt
