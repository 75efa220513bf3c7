# Sweep at 1M rows on one box (run on the GPU box): pause unit (TKSPMV_PACE, 128 cycles) x paced eighths of the field (TKSPMV_PACE_LEVELS)
cd $GRAFT_REPO_ROOT
run() { echo -n "$*: "; env "$@" timeout -k 10 200 python tools/ablate_probe.py ${ROWS:-1000000} 1024 20 none 2>&1 | grep flags | head -1 | cut -c1-75; }
for r in 1 2; do
  run TKSPMV_SMALL_PACKETS=0
  for c in "4 3" "3 3" "5 3" "3 4" "2 4" "4 2" "6 2" "2 5" "8 1" "2 6"; do set -- $c; run TKSPMV_PACE=$1 TKSPMV_PACE_LEVELS=$2; done
done
