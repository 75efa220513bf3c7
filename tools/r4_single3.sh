#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r4j
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests/test_gpu_single.py -x -q -m gpu > "$OUT/tests.log" 2>&1
rc=$?
tail -3 "$OUT/tests.log"
[ $rc -ne 0 ] && { tail -40 "$OUT/tests.log"; exit $rc; }
timeout -k 10 300 python3 tools/r4_single_trace.py > "$OUT/trace.log" 2>&1
grep -E "tkspmv_run device|loop done   |record delivered|selection:" "$OUT/trace.log" | head -13
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$REPO/tools/single_probe.py" --plain 200 > "$OUT/plain.log" 2> "$OUT/plain.err"
f=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/single_kernel_stats.csv"; rm -rf "$OUT/kt"
grep single_kernel "$OUT/single_kernel_stats.csv" | cut -d, -f2-8
cd "$REPO"
timeout -k 10 200 python3 tools/host_path_probe.py > "$OUT/host_path.log" 2>&1; tail -5 "$OUT/host_path.log"
