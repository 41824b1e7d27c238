set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_a; mkdir -p $O
python -m pytest tests/test_fmt_tables.py tests/test_fmt_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/t_fmt.log
python -m pytest tests/test_configs_gpu.py -m gpu -q -s -k "config2" 2>&1 | grep -v "^$" | tail -25 > $O/t_cfg.log
python -m pytest tests/test_variants_gpu.py -m gpu -x -q -k fmt 2>&1 | tail -8 > $O/t_var.log
for i in 1 2; do
for h in 1 0; do FLOAT_FMT_HOIST=$h python tools/probes/fmtbench.py 2>&1 | tail -1; done
done > $O/ab.log
for z in 1 2 4 5 8; do FLOAT_FMT_ZGROUP=$z python tools/probes/fmtbench.py 2>&1 | tail -1; done >> $O/ab.log
FMT_DTYPE=fp16 python tools/probes/fmtbench.py 2>&1 | tail -1 >> $O/ab.log
FLOAT_FMT_TOUCH=0 python tools/probes/fmtbench.py 2>&1 | tail -1 >> $O/ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o fmt -- python3 tools/probes/fmtbench.py > $O/stats.log 2>&1
find $O/stats -name "*kernel_trace.csv" -delete
cat $O/t_fmt.log $O/t_cfg.log $O/t_var.log $O/ab.log
