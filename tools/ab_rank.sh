# A/B at 1M rows on one box (run on the GPU box): round 3's defaults (workgroup-local thresholds, carried, pacing by rank at 2 units)
# against other pauses (TKSPMV_PACE, units of 256 cycles per level) and against round 2's behaviour (TKSPMV_SMALL_PACKETS=0)
cd $GRAFT_REPO_ROOT
run() { echo -n "$*: "; env "$@" timeout -k 10 200 python tools/ablate_probe.py ${ROWS:-1000000} 1024 20 none 2>&1 | grep flags | head -1 | cut -c1-75; }
for r in 1 2 3; do
  run TKSPMV_SMALL_PACKETS=0
  run TKSPMV_DEFAULTS=1
  run TKSPMV_PACE=3
done
