#!/bin/sh
# Per-kernel durations of the FMT step chain under rocprofv3 (graph replay, 250 evaluations x (2 warm-up + 1) runs):
#   sh tools/fmt_trace.sh <tag> [VAR=value ...]     -> gpurun_out/<tag>/summary.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
O=gpurun_out/$tag; mkdir -p $O
FMT_DTYPE=${FMT_DTYPE:-fp16} FMT_REPS=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 tools/probes/fmtbench.py > $O/log.txt 2>&1
tail -1 $O/log.txt
python tools/summarize_trace.py $O/trace fmt_ > $O/summary.csv
python tools/trace_gaps.py $O/trace ${NLAST:-12000} > $O/gaps.csv
find $O/trace -name "*kernel_trace.csv" -delete
head -${TOPN:-16} $O/gaps.csv | cut -c1-200
