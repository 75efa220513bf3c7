#!/bin/bash
# The driver's command line in several trees in turn on ONE box (trees built by tools/ab_variants.sh build; `.` = this one):
#   bash tools/ab_bench_line.sh ROUNDS TREE...   -> one line per run: tree, kernel_us, fraction, the period tkspmv_create measured
rounds=$1; shift
for r in $(seq 1 "$rounds"); do
    for v in "$@"; do
        d=_ab/$v; [ "$v" = . ] && d=.
        (cd "$d" && python3 bench.py --gpus 1 --steps 20 --warmup 5 --skip-warm --cpu-seconds 0 --traffic off 2>/dev/null | grep "^{" | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(json.dumps({'tree': '$v', 'kernel_us': round(r['kernel_us'], 2), 'frac': round(r['frac'], 3), 'period_ns': r.get('pace_period_ns'), 'read_only_us': round(r.get('read_only_us') or 0, 2), 'read_only_paced_us': round(r.get('read_only_paced_us') or 0, 2), 'checks_failed': r.get('checks_failed'), 'tune_launches': r.get('launches_of_the_pacing_measurement')}))") || exit 1
    done
done
