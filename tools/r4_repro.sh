#!/bin/bash
echo "== main"; timeout -k 5 100 python3 tools/r4_repro.py 1000000 2>&1 | grep -E "time_queries|fault|done|info" | cut -c1-220
exit 0
