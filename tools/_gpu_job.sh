cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2r
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2r/tests.log 2>&1; tail -4 gpurun_out/r2r/tests.log
timeout 600 python bench.py --no-cpu-baseline --passes default --no-train > gpurun_out/r2r/bench.json 2> gpurun_out/r2r/bench.err; tail -3 gpurun_out/r2r/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r2r/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
