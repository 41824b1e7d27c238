cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 0 4 8 12 16; do echo "tail $t: $(FLOAT_DEC_RIDE_TAIL=$t python tools/probes/dec_host.py 2>&1 | tail -1)"; done
python -m pytest tests/test_dec_gpu.py -m gpu -x -q 2>&1 | tail -2
