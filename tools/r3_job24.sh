#!/bin/bash
TAG=${1:-r3aa}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=300 > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?"; tail -4 $OUT/tests_gpu.log
timeout 900 python tools/abn.py --rounds 2 base=- shade_fp32=-,ARAH_SHADE_ENGINE=fp32 2>&1 | tee $OUT/abn.txt
timeout 600 python bench.py --steps 6 --warmup 2 --no-train > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "psnr_vs_oracle_db", "mask_agreement", "value_full_shading", "value_strict")})
print("one frame", d.get("one_frame_at_a_time"))
print("roofline", d.get("roofline", {}).get("frac"), d.get("roofline_k_density", {}).get("frac"))
PY
