#!/bin/bash
set -u
for r in 1 2; do for d in . _ab/nogate _ab/r3; do (cd $d && TKSPMV_LOCAL=0 timeout -k 10 120 python3 /tmp/exact.py 1000000 2>&1 | grep rows) || exit 1; done; done
