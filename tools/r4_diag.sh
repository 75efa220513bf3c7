#!/bin/bash
# Round-4 diagnostics of single launches (gpurun): kernel durations by rocprofv3 + per-wave timelines.
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r4a
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$REPO/tools/r4_diag.py" 1000000 100 > "$OUT/diag.log" 2> "$OUT/diag.err"
TKSPMV_DBG_FLAGS=2 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_nooffer" -- python3 "$REPO/tools/r4_diag.py" 1000000 100 fused,batch1 > "$OUT/diag_nooffer.log" 2> "$OUT/diag_nooffer.err"
cd "$REPO"
NQ=1 timeout -k 10 200 python3 tools/batch_trace.py > "$OUT/trace_batch1.log" 2>&1
timeout -k 10 300 python3 tools/single_probe.py > "$OUT/single_probe.log" 2>&1
for d in kt kt_nooffer; do
  f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${d}_kernel_stats.csv"
done
rm -rf "$OUT/kt" "$OUT/kt_nooffer"
tail -5 "$OUT/diag.log"
