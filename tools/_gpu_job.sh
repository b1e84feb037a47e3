cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2x
for b in 0.003 0.01 0.03; do
timeout 600 python bench.py --beta $b --no-cpu-baseline --no-train --steps 4 --warmup 1 > gpurun_out/r2x/bench_beta_$b.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r2x/bench_beta_$b.json"))
print("beta $b: default %.0f rays/s %.1f ms | full %.0f | exact %.0f | strict %.0f | n_col/ray %.2f of n_density/ray %.2f" % (d["value"], d["ms_per_step"], d["full_shading"]["value"], d["exact_fp32_engine"]["value"], d["strict"]["value"], d["work"]["per_ray"]["n_col"], d["work"]["per_ray"]["n_density"]))
PY
done
