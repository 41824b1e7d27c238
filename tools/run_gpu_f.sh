cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_enc_gpu.py tests/test_dec_gpu.py tests/test_nodes_gpu.py tests/test_nodes_va_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -3
