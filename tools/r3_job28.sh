#!/bin/bash
TAG=${1:-r3ae}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-train --passes default > $OUT/a.json 2> $OUT/a.err; echo "bench streams1 rc=$?"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --passes default > $OUT/b.json 2> $OUT/b.err; echo "bench default streams rc=$?"
