#!/bin/bash
OUT=gpurun_out/r4o; mkdir -p $OUT
ROWS=125000 NQ=32 timeout -k 5 120 python3 tools/batch_trace.py > $OUT/new.log 2>&1
(cd _ab/r3 && ROWS=125000 NQ=32 timeout -k 5 120 python3 tools/batch_trace.py > ../../$OUT/r3.log 2>&1)
for f in r3 new; do echo "== $f"; grep -E "selector stamps|selection of|waves  q|server q|qM |gap qM|spread" $OUT/$f.log | cut -c1-200; done
