cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "--- $*"; env "$@" python tools/probes/e2e_phases.py 2>&1 | grep -v amdgpu.ids | head -3; }
run DUMMY_SIDE=wait MODE=none
run SIDE_COPY=wait MODE=none
run SIDE_COPY=sync MODE=none
run FULL_SYNC=1 MODE=pipe
run GPU_MAX_HW_QUEUES=8 MODE=pipe
run MODE=serial
