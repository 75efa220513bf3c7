#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r4l
rm -rf "$OUT"; mkdir -p "$OUT"
for t in 0 4 0 4; do
  TKSPMV_SINGLE_TUNE=$t timeout -k 10 300 python3 tools/r4_single_trace.py > "$OUT/trace_tune$t.log" 2>&1
  echo "== tune $t"; grep -E "tkspmv_run device|loop done   |record delivered|selection:" "$OUT/trace_tune$t.log" | sed -n 5,13p
done
cd /tmp && export TMPDIR=/tmp
for t in 0 4; do
TKSPMV_SINGLE_TUNE=$t timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt$t" -- python3 "$REPO/tools/single_probe.py" --plain 200 > "$OUT/plain$t.log" 2> "$OUT/plain$t.err"
f=$(find "$OUT/kt$t" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/single_kernel_stats_tune$t.csv"; rm -rf "$OUT/kt$t"
echo "== tune $t"; grep single_kernel "$OUT/single_kernel_stats_tune$t.csv" | cut -d, -f2-8
done
