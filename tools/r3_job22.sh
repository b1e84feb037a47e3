#!/bin/bash
TAG=${1:-r3y}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 400 python tools/stress_streams.py --runs 200 --frames 8 --streams 3 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tee $OUT/streams_soak.txt
echo "soak rc=${PIPESTATUS[0]}"
