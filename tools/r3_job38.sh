#!/bin/bash
# BASELINE.json configs 1 and 5 through bench.py on the final build
TAG=${1:-r3ar}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python bench.py --size 256 --n-steps 32 --steps 8 --warmup 2 --no-train --no-cpu-baseline --passes default > $OUT/config1.json 2> $OUT/config1.err; echo "config1 rc=$?"
timeout 900 python bench.py --size 1024 --n-steps 128 --config h36m --steps 3 --warmup 1 --no-train --no-cpu-baseline --passes default > $OUT/config5.json 2> $OUT/config5.err; echo "config5 rc=$?"
python - <<PY
import json
for f in ("config1", "config5"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["value"]), round(d["ms_per_step"], 2), "one at a time", round(d["one_frame_at_a_time"]["value"]), round(d["one_frame_at_a_time"]["ms_per_step"], 2))
    except Exception as e:
        print(f, "failed", e)
PY
