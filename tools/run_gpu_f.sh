cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_variants_gpu.py -m gpu -x -q -k conv2_flow 2>&1 | tail -3
