cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fmt_gpu.py tests/test_edge_cases_gpu.py tests/test_fmt_fp32_gpu.py tests/test_configs_gpu.py tests/test_variants_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do python -m pytest tests/test_fmt_gpu.py -m gpu -x -q -k capturable 2>&1 | tail -1; done
FMT_DTYPE=fp16 python tools/probes/fmtbench.py 2>&1 | tail -1 | cut -c1-70
