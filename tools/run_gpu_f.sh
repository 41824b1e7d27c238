cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_aud_gpu.py tests/test_configs_gpu.py -m gpu -x -q -s -k "long_audio or sixty or golden" 2>&1 | grep -v "^$" | tail -12
