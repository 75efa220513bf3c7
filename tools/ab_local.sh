# A/B of the small-matrix settings of the batch kernel over shard sizes, on one box (run on the GPU box):
# selector workgroups (TKSPMV_SELECTORS), partition lengths (TKSPMV_MIN_PACKETS), workgroup-local thresholds (TKSPMV_LOCAL)
set -e
cd $GRAFT_REPO_ROOT
run() { echo "$*"; env SWEEP=small "$@" timeout -k 10 200 python tools/size_sweep.py 2>&1 | grep F32; }
run TKSPMV_SMALL_PACKETS=0
run TKSPMV_DEFAULTS=1
run TKSPMV_LOCAL=1
run TKSPMV_LOCAL=2
run TKSPMV_LOCAL=0
run TKSPMV_SMALL_PACKETS=45000
