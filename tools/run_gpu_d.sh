cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_d; mkdir -p $O
BATCHES=1,2,4 python tools/probes/fmtbatch.py 2>&1 | grep "B=" | tee $O/sweep.log
for plan in 3,4,4 3,4,8 5,4,4 6,2,8 6,2,4 5,2,8 3,2,8 4,2,8; do
  echo "plan $plan" | tee -a $O/sweep.log
  FLOAT_FMT_PLAN=$plan BATCHES=2,4 python tools/probes/fmtbatch.py 2>&1 | grep "B=" | tee -a $O/sweep.log
done
