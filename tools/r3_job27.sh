#!/bin/bash
TAG=${1:-r3ad}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 300 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex run -ex "bt 40" -ex "info threads" --args python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-train --passes default > $OUT/gdb.log 2>&1
echo "gdb rc=$?"
grep -n "terminate\|SIGABRT\|#[0-9]" $OUT/gdb.log | head -60
