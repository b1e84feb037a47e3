cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 200 python bench.py --size 1024 --n-steps 128 --config h36m --steps 3 --warmup 1 --no-cpu-baseline --passes default --no-train > gpurun_out/r2g/bench_config5.json 2>/dev/null
timeout 120 python bench.py --size 256 --n-steps 32 --steps 8 --warmup 2 --no-cpu-baseline --passes default --no-train > gpurun_out/r2g/bench_config1.json 2>/dev/null
python - <<PY
import json
for c in ("5","1"):
    d=json.load(open("gpurun_out/r2g/bench_config%s.json" % c)); print(c, d["value"], d["ms_per_step"], d.get("frames_in_flight",{}).get("value"), d.get("frames_in_flight",{}).get("ms_per_step"))
PY
