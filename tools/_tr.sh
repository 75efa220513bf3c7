cd $GRAFT_REPO_ROOT
for cfg in "50000 2" "50000 1" "125000 2" "250000 2"; do
  set -- $cfg
  echo "== rows $1 min_packets $2"
  ROWS=$1 NQ=8 TKSPMV_MIN_PACKETS=$2 timeout -k 10 200 python tools/batch_trace.py 2>&1 | grep -E "selector stamps|selection of|qM duration \(|gap qM|qM start ->|waves  qM|server q1"
done
