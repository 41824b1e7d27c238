#!/bin/sh
python -m pytest tests/test_variants_gpu.py tests/test_fmt_gpu.py tests/test_edge_cases_gpu.py tests/test_configs_gpu.py tests/test_fmt_fp32_gpu.py -x -q -m gpu 2>&1 | tail -4
BATCHES=1,2,4 python tools/probes/fmtbatch.py 2>&1 | grep "B="
python bench.py 2>/dev/null | cut -c1-330
