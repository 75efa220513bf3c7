#!/bin/bash
# Pacing sweep (PACE quantum x PACE_LEVELS) of one or more builds of the library on ONE box: tools/pace_sweep.sh OUT LIB...
# (LIB = . for the tree's own library, or a directory under _ab/)
out=$1; shift
for lib in "$@"; do
    so=$PWD/approximate-spmv-topk_amd/libtkspmv.so; [ "$lib" != . ] && so=$PWD/_ab/$lib/approximate-spmv-topk_amd/libtkspmv.so
    for levels in ${LEVELS:-3 4 6}; do
        for pace in ${PACES:-2 4 5 6 8}; do
            TKSPMV_LIB=$so TKSPMV_PACE=$pace TKSPMV_PACE_LEVELS=$levels timeout -k 10 120 python tools/ladder_probe.py "${lib}_p${pace}_l${levels}" >> "$out"
        done
    done
done
python - "$out" <<PY
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(f"{d['name']:24s} sustained {d['sustained_median_us']:6.2f} (p95/med {d['p95_over_median']:.3f})  driver line {d['driver_line_kernel_us']:6.2f}  floor {d['read_only_us']:6.2f}")
PY
