#!/bin/bash
# One GPU-box session: GPU parity tests, the default bench line, a rocprofv3 kernel-trace summary of the default
# path.  Usage (from the repo root, on the GPU box):  bash tools/gpu_round.sh <tag> [pmc]
# Everything lands under gpurun_out/<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests -m gpu -x -q --timeout=240 > $OUT/tests.log 2>&1
echo "pytest rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
timeout 600 python bench.py --steps 8 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_default.json"))
    print("rays/s", d["value"], "ms", d["ms_per_step"], "canon frac", d["roofline"]["frac"], "canon_ms", d["roofline"]["avg_launch_ms"],
          "density frac", d["roofline_k_density"]["frac"], "dens_ms", d["roofline_k_density"]["avg_launch_ms"])
    print("vs_baseline", d.get("vs_baseline"), (d.get("gpu_torch_baseline") or {}).get("value"), "tiers", d.get("tiers", {}).get("samples_per_ray"))
    for k in ("untiered", "full_shading", "exact_fp32_engine", "strict"):
        if k in d: print(k, d[k]["value"], d[k]["ms_per_step"])
    print("one frame at a time", d.get("one_frame_at_a_time"))
    print("training", d.get("training"))
    print("psnr", d.get("psnr_vs_oracle_db"), "mask", d.get("mask_agreement"), "work", d["work"]["per_ray"])
except Exception as e:
    print("bench parse failed", e)
PY
fi
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-gpu-baseline --no-train --passes default > $OUT/bench_prof.json 2> $OUT/prof.err
cd $ROOT
DB=$(ls -S $(find $OUT/prof -name "*.db") | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt && head -30 $OUT/kernel_stats.txt
[ -n "$DB" ] && python tools/rocpd_timeline.py $DB --all > $OUT/timeline.txt
if [ "$2" = "pmc" ]; then
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
    N=$(echo $C | cut -d' ' -f1)
    timeout 900 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$N -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-gpu-baseline --no-train --passes default > $OUT/pmc_$N.log 2>&1
  done
  # the shade-everything path (the reference's amount of work): its own FETCH / WRITE passes
  for C in FETCH_SIZE WRITE_SIZE; do
    ARAH_FULL_SHADING=1 timeout 900 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmcfull_$C -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-gpu-baseline --no-train --passes default > $OUT/pmcfull_$C.log 2>&1
  done
  ARAH_FULL_SHADING=1 timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $OUT/pmcfull_SQ -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-gpu-baseline --no-train --passes default > $OUT/pmcfull_SQ.log 2>&1
  cd $ROOT
  big() { ls -S $(find $1 -name "*.db") 2>/dev/null | head -1; }
  F=$(big $OUT/pmc_FETCH_SIZE); W=$(big $OUT/pmc_WRITE_SIZE)
  [ -n "$F" ] && [ -n "$W" ] && python tools/rocpd_pmc.py $F $W > $OUT/pmc_traffic.json
  F=$(big $OUT/pmcfull_FETCH_SIZE); W=$(big $OUT/pmcfull_WRITE_SIZE)
  [ -n "$F" ] && [ -n "$W" ] && python tools/rocpd_pmc.py $F $W > $OUT/pmc_traffic_full.json
  S=$(big $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES)
  [ -n "$S" ] && python tools/rocpd_sq.py $S > $OUT/pmc_sq.json
  S=$(big $OUT/pmcfull_SQ)
  [ -n "$S" ] && python tools/rocpd_sq.py $S > $OUT/pmc_sq_full.json
  # the databases are large: keep the summaries only
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES $OUT/pmcfull_FETCH_SIZE $OUT/pmcfull_WRITE_SIZE $OUT/pmcfull_SQ
fi
rm -rf $OUT/prof
