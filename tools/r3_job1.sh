#!/bin/bash
# round 3, GPU session 1: point-owning-wave loop C -- parity first, then A/B against round 2's tile kernel
TAG=${1:-r3a}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q --timeout=240 -k "broyden3 or tracer_against or forward_against or split_engine_matches" > $OUT/tests_canon.log 2>&1
echo "canon tests rc=$?" | tee -a $OUT/tests_canon.log
tail -15 $OUT/tests_canon.log
for K in tile wave wave_l2 wave tile; do
  ARAH_CANON_KERNEL=$K timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-train --passes default > $OUT/bench_$K.json 2> $OUT/bench_$K.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$K.json"))
    print("$K", "rays/s %.3g" % d["value"], "ms %.2f" % d["ms_per_step"], "canon_ms %.2f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"],
          "dens_ms %.2f" % d["roofline_k_density"]["avg_launch_ms"], "psnr", d.get("psnr_vs_oracle_db"), "mask", d.get("mask_agreement"), "evals", d["roofline"]["evaluations_per_launch"])
except Exception as e:
    print("$K bench parse failed", e)
PY
done
python tools/phase_clocks.py run 3 > $OUT/phase_clocks_wave.txt 2>&1; cat $OUT/phase_clocks_wave.txt | tail -12
timeout 900 python -m pytest tests -m gpu -x -q --timeout=240 > $OUT/tests.log 2>&1
echo "pytest rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-train --passes default > $OUT/bench_prof.json 2> $OUT/prof.err
cd $ROOT
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt && head -14 $OUT/kernel_stats.txt
[ -n "$DB" ] && python tools/rocpd_timeline.py $DB --all > $OUT/timeline.txt
rm -rf $OUT/prof
