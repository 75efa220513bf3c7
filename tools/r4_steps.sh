#!/bin/bash
echo "== new"; timeout -k 5 200 python3 tools/steps_probe.py 2>&1 | grep -E "n= +(1|20|32|128):|fit|cold" | cut -c1-120
echo "== r3"; (cd _ab/r3 && timeout -k 5 200 python3 tools/steps_probe.py 2>&1 | grep -E "n= +(1|20|32|128):|fit|cold" | cut -c1-120)
bash tools/ab_variants.sh run 1 125000 1024 20 r3 . 2>&1 | grep flags | cut -c1-100
TKSPMV_LOCAL=0 bash tools/ab_variants.sh run 1 1000000 1024 20 r3 . 2>&1 | grep flags | cut -c1-100
bash tools/ab_variants.sh run 1 3000000 1024 20 r3 . 2>&1 | grep flags | cut -c1-100
