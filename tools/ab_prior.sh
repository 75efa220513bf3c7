# A/B of the batch kernel's prior thresholds (TKSPMV_PRIOR / TKSPMV_PRIOR_BETA / TKSPMV_PRIOR_RISE) on one box (run on the GPU box)
set -e
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for c in "0 0" "0.6 1e9" "0.6 1.02" "0.7 1.02" "0.8 1.02" "0.65 1e9"; do
    set -- $c
    echo -n "beta=$1 rise=$2: "; TKSPMV_PRIOR=$([ $1 = 0 ] && echo 0 || echo 1) TKSPMV_PRIOR_BETA=$1 TKSPMV_PRIOR_RISE=$2 timeout -k 10 200 python tools/ablate_probe.py 1000000 1024 20 none 2>&1 | grep flags | head -1 | cut -c1-70
  done
done
