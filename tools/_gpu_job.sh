cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
( time timeout 250 python bench.py > gpurun_out/r2e/bench_default.json 2> gpurun_out/r2e/bench.err ) 2>&1 | tail -3; echo "rc=$?"
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2e/bench_default.json"))
    print("value", d["value"], d["ms_per_step"], "in flight", d.get("frames_in_flight"), "full", d["full_shading"]["value"], "exact", d["exact_fp32_engine"]["value"], "strict", d["strict"]["value"], "train", d["training"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"])
    print(d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline_k_density"]["frac"], d["roofline_k_density"]["avg_launch_ms"])
except Exception as e:
    print("parse failed", e); print(open("gpurun_out/r2e/bench.err").read()[-500:])
PY
