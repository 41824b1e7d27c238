set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_b; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cat $O/bench.json
FLOAT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench2.json 2> $O/bench2.err; echo rc=$?; tail -3 $O/bench2.err; cat $O/bench2.json
FLOAT_BENCH_BACKEND=gloo python bench.py --gpus 2 --mode window --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $O/bench2w.json 2> $O/bench2w.err; echo rc=$?; tail -3 $O/bench2w.err; cat $O/bench2w.json
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/t_all.log; cat $O/t_all.log
