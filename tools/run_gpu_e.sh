cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fmt_fp32_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -30
python -m pytest tests/test_fmt_gpu.py tests/test_fmt_tables.py tests/test_aud_gpu.py -m gpu -x -q 2>&1 | tail -3
