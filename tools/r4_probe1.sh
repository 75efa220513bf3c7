#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r4b
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 "$REPO/tools/r4_probe1.py" > "$OUT/probe.log" 2> "$OUT/probe.err"
cd "$REPO"
f=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "$OUT/probe_kernels.txt" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "read_probe" in n:
        d[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for n, v in d.items():
    v = sorted(v)
    short = [x for x in v if x < 200000]
    print(n[:60], "launches", len(v), "one-pass launches: n", len(short), "min", short[0] if short else None, "median", short[len(short)//2] if short else None)
PY
rm -rf "$OUT/kt"
cat "$OUT/probe.log" "$OUT/probe_kernels.txt"
