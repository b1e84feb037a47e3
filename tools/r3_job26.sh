#!/bin/bash
TAG=${1:-r3ac}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
python tools/_exit_probe.py > $OUT/p1.log 2>&1; echo "probe default rc=$?"
ARAH_SHADE_ENGINE=fp32 python tools/_exit_probe.py > $OUT/p2.log 2>&1; echo "probe shade fp32 rc=$?"
ARAH_EARLY_BODY_TABLES=0 python tools/_exit_probe.py > $OUT/p3.log 2>&1; echo "probe inline body rc=$?"
ARAH_SHADE_ENGINE=fp32 ARAH_EARLY_BODY_TABLES=0 python tools/_exit_probe.py > $OUT/p4.log 2>&1; echo "probe both off rc=$?"
python -X faulthandler bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-train --passes default > $OUT/f.json 2> $OUT/f.err; echo "bench faulthandler rc=$?"
grep -v "amdgpu.ids" $OUT/f.err | head -40
