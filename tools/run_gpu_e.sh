cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_e; mkdir -p $O
MODE=inorder python tools/probes/e2e_phases.py 2>&1 | grep -v amdgpu.ids | head -3
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-1400
python -m pytest tests/test_dec_gpu.py tests/test_pipeline_gpu.py tests/test_nodes_gpu.py tests/test_nodes_va_gpu.py tests/test_variants_gpu.py -m gpu -x -q 2>&1 | tail -3
