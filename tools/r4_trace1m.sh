#!/bin/bash
OUT=gpurun_out/r4q; mkdir -p $OUT
ROWS=1000000 NQ=20 timeout -k 5 120 python3 tools/batch_trace.py > $OUT/new.log 2>&1
(cd _ab/r3 && ROWS=1000000 NQ=20 timeout -k 5 120 python3 tools/batch_trace.py > ../../$OUT/r3.log 2>&1)
for f in r3 new; do echo "== $f"; grep -E "selector stamps|selection of|waves  |server q|qM |gap qM" $OUT/$f.log | cut -c1-220; done
