#!/bin/sh
# decoder: channel-block order A/B (FLOAT_DEC_CB_ORDER), then the decoder parity tests on the new order
for o in 0 1 0 1; do
  echo "== FLOAT_DEC_CB_ORDER=$o"; FLOAT_DEC_CB_ORDER=$o DEC_MAXF=32 python tools/probes/decbench.py 2>&1 | grep decode
done
python -m pytest tests/test_dec_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -4
