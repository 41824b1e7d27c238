cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 2 4 5 2 4; do echo "variant $v: $(FLOAT_FMT_WIDE_VARIANT=$v FMT_DTYPE=fp16 FMT_SAVE=/tmp/v$v.pt python tools/probes/fmtbench.py 2>&1 | tail -1 | cut -c1-75)"; done
python -c "
import torch
a=torch.load('/tmp/v2.pt'); b=torch.load('/tmp/v4.pt'); c=torch.load('/tmp/v5.pt'); print('bitwise 2==4', torch.equal(a,b), '2==5', torch.equal(a,c))"
for z in 2 4 8; do echo "variant 4 zgroup $z: $(FLOAT_FMT_WIDE_VARIANT=4 FLOAT_FMT_ZGROUP=$z FMT_DTYPE=fp16 python tools/probes/fmtbench.py 2>&1 | tail -1 | cut -c1-75)"; done
