#!/bin/bash
for i in 1 2 3; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --skip-warm --cpu-seconds 0 --traffic off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=d['timing']
print('steps20: kernel_us %.2f frac %.3f | sustained median %.2f p95 %.2f | with gaps median %.2f | read-only %.2f | nonstat %.2f fails %d' % (r['kernel_us'], r['frac'], t['kernel_us_median'], t['kernel_us_p95'], t['one_call_per_repetition']['kernel_us_median'], r['read_only']['us_per_pass'], d['nonstationary']['kernel_us_median'], d['nonstationary']['checks_failed']))"
(cd _ab/r3 && python3 bench.py --gpus 1 --steps 20 --warmup 5 --skip-warm --cpu-seconds 0 --traffic off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=d['timing']
print('r3 steps20: kernel_us %.2f frac %.3f | reps median %.2f p95 %.2f | read-only %.2f' % (r['kernel_us'], r['frac'], t['kernel_us_median'], t['kernel_us_p95'], r['read_only']['us_per_pass']))")
done
