#!/bin/bash
TAG=${1:-r3ab}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
python bench.py --steps 3 --warmup 1 --streams 3 --no-cpu-baseline --no-train --passes default > $OUT/a.json 2> $OUT/a.err; echo "A streams3 rc=$?"
python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-train --passes default > $OUT/b.json 2> $OUT/b.err; echo "B streams1 rc=$?"
ARAH_EARLY_BODY_TABLES=0 python bench.py --steps 3 --warmup 1 --streams 3 --no-cpu-baseline --no-train --passes default > $OUT/c.json 2> $OUT/c.err; echo "C streams3 inline-body rc=$?"
python bench.py --steps 3 --warmup 1 --streams 3 --no-cpu-baseline --no-train --passes all > $OUT/d.json 2> $OUT/d.err; echo "D streams3 all passes rc=$?"
python bench.py --steps 3 --warmup 1 --streams 3 --no-train --passes default > $OUT/e.json 2> $OUT/e.err; echo "E streams3 with cpu baseline rc=$?"
tail -2 $OUT/*.err | cut -c1-200
