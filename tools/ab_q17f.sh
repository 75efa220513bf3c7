# A/B of BASELINE configs[4]'s kernel (row-per-lane byte stream, one query per pass) between builds on one box: tools/ab_q17f.sh NAME...
cd "$(dirname "$0")/.."
for r in 1 2; do
  for v in "$@"; do
    d=_ab/$v; [ "$v" = . ] && d=.
    echo -n "$v: "; (cd $d && timeout -k 10 200 python tools/q17f_probe.py --only 2>&1 | grep "row per lane" | sed 's/.*MB  *\([0-9.]* us\/query\).*/\1/')
  done
done
