# One box (run on the GPU box): other value types and shapes under workgroup-local thresholds + pacing
cd $GRAFT_REPO_ROOT
run() { echo -n "$*: "; env SWEEP=wide "$@" timeout -k 10 300 python tools/size_sweep.py 2>&1 | grep -E "F32|F16|Q1_7" | tr '\n' ' '; echo; }
for p in off 1 2 3; do
  if [ $p = off ]; then E="TKSPMV_SMALL_PACKETS=0"; else E="TKSPMV_SMALL_PACKETS=400000 TKSPMV_PACE=$p"; fi
  run $E SWEEP_ROWS=1000000 SWEEP_PREC=F16
  run $E SWEEP_ROWS=1000000 SWEEP_COLS=512 SWEEP_NNZ=40 SWEEP_PREC=Q1_7
  run $E SWEEP_ROWS=1000000 SWEEP_COLS=512 SWEEP_NNZ=40 SWEEP_PREC=F32
  run $E SWEEP_ROWS=1000000 SWEEP_PREC=FIXED
  run $E SWEEP_ROWS=1000000 SWEEP_K=8
done
