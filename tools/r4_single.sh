#!/bin/bash
# Round 4: the single-query kernel -- parity first, then its duration by rocprofv3 and its per-wave timeline.
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r4c
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests/test_gpu_single.py -x -q -m gpu -s > "$OUT/tests.log" 2>&1
rc=$?
tail -15 "$OUT/tests.log"
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$REPO/tools/single_probe.py" --plain 200 > "$OUT/plain.log" 2> "$OUT/plain.err"
cd "$REPO"
f=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/single_kernel_stats.csv"; rm -rf "$OUT/kt"
head -4 "$OUT/single_kernel_stats.csv" | cut -c1-160
timeout -k 10 200 python3 tools/r4_diag.py 1000000 100 fused > "$OUT/diag.log" 2>&1; cat "$OUT/diag.log"
timeout -k 10 200 python3 tools/host_path_probe.py > "$OUT/host_path.log" 2>&1; tail -6 "$OUT/host_path.log"
