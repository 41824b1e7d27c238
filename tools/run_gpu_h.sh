#!/bin/sh
# large-tile chain GEMMs for stacked clips: parity of the batched sampler, then throughput A/B and per-layer plans
mkdir -p gpurun_out/big
python -m pytest tests/test_edge_cases_gpu.py -x -q -m gpu -k "batch" 2>&1 | tail -5
for cfg in "0" "320"; do
  echo "== FLOAT_FMT_BIG_ROWS=$cfg"
  FLOAT_FMT_BIG_ROWS=$cfg BATCHES=1,2,4 python tools/probes/fmtbatch.py 2>&1 | grep "B="
done
for plan in "12,1 12,4 12,1 12,4" "6,1 6,2 6,1 6,4" "6,1 12,4 6,1 12,4" "12,1 6,4 12,1 6,4" "6,1 6,4 6,1 6,2"; do
  set -- $plan
  echo "== qkv $1 proj $2 fc1 $3 fc2 $4"
  FLOAT_FMT_BIG_QKV=$1 FLOAT_FMT_BIG_PROJ=$2 FLOAT_FMT_BIG_FC1=$3 FLOAT_FMT_BIG_FC2=$4 BATCHES=2,4 python tools/probes/fmtbatch.py 2>&1 | grep "B="
done
