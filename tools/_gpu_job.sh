cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2g/tests.log 2>&1; tail -4 gpurun_out/r2g/tests.log
timeout 300 python tools/phase_clocks.py run 3 > gpurun_out/r2g/clocks.txt 2>&1; cat gpurun_out/r2g/clocks.txt
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --passes default > gpurun_out/r2g/bench.json 2>gpurun_out/r2g/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r2g/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
