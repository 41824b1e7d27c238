cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_final
python bench.py > gpurun_out/r02_final/bench.json 2> gpurun_out/r02_final/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r02_final/bench.json')); print(d['value'], d['ms_per_step']); print(d['roofline']); print(d.get('warnings'))"
