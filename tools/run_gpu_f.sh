cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
b() { echo "$*: $(env "$@" FMT_DTYPE=fp16 python tools/probes/fmtbench.py 2>&1 | tail -1 | cut -c24-60)"; }
b X=1
for p in 3,4,8 3,2,8 3,2,16 4,2,8 6,2,8 3,4,16 5,2,8 5,4,8 2,4,8 4,1,8 3,4,4; do b FLOAT_FMT_PLAN_FC1=$p; done
for p in 3,4,8 3,2,8 4,2,8 6,2,8 3,4,16 3,2,16 5,4,8 3,4,4; do b FLOAT_FMT_PLAN_FC2=$p; done
b X=2
