#!/bin/bash
# tools/scale.sh [outdir] : the scaling runs of BASELINE.json on ONE node with up to 8 MI355X, one process per GPU over RCCL.
#   inference (configs[1] / [3]): bench.py --gpus N for N = 1, 2, 4, 8 -- frames sharded frame i -> rank i mod N, no data-path
#                                 collective, weak scaling; one JSON line per N in <outdir>/bench_nN.json
#   training  (configs[2])      : tools/train_bench.py --exchange native (GradientExchange: bucketed all-reduce overlapped with
#                                 backward) and --exchange ddp (torch DDP) at N = 2, 4, 8 -> <outdir>/train_<exchange>_nN.json
#   the two RCCL tests that skip on a one-GPU box
# Rendezvous on 127.0.0.1; HSA_ENABLE_IPC_MODE_LEGACY=0 for dmabuf IPC.  N larger than the visible GPU count is skipped.
OUT=${1:-gpurun_out/scale}
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $NGPU"
PORT=29511
for N in 1 2 4 8; do
  [ $N -gt $NGPU ] && { echo "skip N=$N"; continue; }
  if [ $N -eq 1 ]; then
    python bench.py --gpus 1 --steps 8 --warmup 2 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $N --steps 8 --warmup 2 > $OUT/bench_n$N.json 2> $OUT/bench_n$N.err
    PORT=$((PORT + 1))
    for EX in native ddp; do
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        tools/train_bench.py --exchange $EX --steps 20 --warmup 3 > $OUT/train_${EX}_n$N.json 2> $OUT/train_${EX}_n$N.err
      PORT=$((PORT + 1))
    done
  fi
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_n$N.json").read().strip().splitlines()[-1])
    print("N=$N  %.3f M rays/s  %.2f ms/frame" % (d["value"] / 1e6, d["ms_per_step"]))
except Exception as e:
    print("N=$N  no line:", e)
PY
done
[ $NGPU -ge 2 ] && python -m pytest tests/test_train_entry.py tests/test_bench_sharding.py -q -m gpu -k rccl 2>&1 | tail -3
