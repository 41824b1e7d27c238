cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 166 230 182 246 167 174 38 134; do
echo "touch $t: $(FLOAT_FMT_TOUCH=$t FMT_DTYPE=fp16 FMT_REPS=4 python tools/probes/fmtbench.py 2>&1 | tail -1 | cut -c1-80)"
done
echo "touch 166 again: $(FLOAT_FMT_TOUCH=166 FMT_DTYPE=fp16 FMT_REPS=4 python tools/probes/fmtbench.py 2>&1 | tail -1 | cut -c1-80)"
