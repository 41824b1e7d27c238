cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_edge_cases_gpu.py -m gpu -x -q -k long_grids 2>&1 | tail -5
