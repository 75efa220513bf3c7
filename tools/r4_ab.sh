#!/bin/bash
OUT=gpurun_out/r4n; mkdir -p $OUT
echo "exact mode (TKSPMV_LOCAL=0)"
TKSPMV_LOCAL=0 bash tools/ab_variants.sh run 2 1000000 1024 20 r3 . 2>&1 | tee $OUT/ab_1m_exact.log
bash tools/ab_variants.sh run 1 3000000 1024 20 r3 . 2>&1 | tee $OUT/ab_3m.log
