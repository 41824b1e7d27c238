cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2_all
python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r2_all/t.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
