# A/B of single fused launches (tkspmv_run's kernel) between builds on ONE box: tools/ab_single.sh ROUNDS NAME... (NAME under _ab/, or .)
cd "$(dirname "$0")/.."
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    d=_ab/$v; [ "$v" = . ] && d=.
    echo -n "$v: "; (cd $d && timeout -k 10 200 python tools/single_probe.py 2>&1 | grep -E "tkspmv_run kernel_ns" | head -1)
  done
done
