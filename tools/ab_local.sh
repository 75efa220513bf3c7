# A/B of the small-matrix settings of the batch kernel over shard sizes, on one box (run on the GPU box):
# selector workgroups (TKSPMV_SELECTORS), partition lengths (TKSPMV_MIN_PACKETS), workgroup-local thresholds (TKSPMV_LOCAL,
# TKSPMV_LOCAL_CARRY, TKSPMV_LOCAL_BETA); TKSPMV_SMALL_PACKETS=0 switches all of them off (round 2's behaviour)
set -e
cd $GRAFT_REPO_ROOT
run() { echo "$*"; env "$@" timeout -k 10 300 python tools/size_sweep.py 2>&1 | grep -E "F32|small matrix"; }
run SWEEP=small TKSPMV_SMALL_PACKETS=0
run SWEEP=small TKSPMV_DEFAULTS=1
run SWEEP=mid TKSPMV_SMALL_PACKETS=0
run SWEEP=mid TKSPMV_DEFAULTS=1
