#!/bin/bash
# A/B runs of kernel variants on ONE box (boxes of the pool differ by several per cent, and so do engine instances):
#   tools/ab_variants.sh build NAME "-DFLAG=..."   copies the tree to _ab/NAME and builds it with the extra flags
#   tools/ab_variants.sh run ROUNDS ROWS COLS NNZ NAME...   (on the GPU box) interleaved runs, production kernel only
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    d=_ab/$2
    rm -rf "$d"; mkdir -p "$d"
    cp -r approximate-spmv-topk_amd include Makefile _pkg.py bench.py tools oracle "$d"/
    mkdir -p "$d/tests"; cp tests/_*.py tests/conftest.py tests/oracle_lib.py "$d/tests/" 2>/dev/null || true
    rm -f "$d"/approximate-spmv-topk_amd/libtkspmv.so
    (cd "$d" && sed -i "s/^HIPFLAGS := /HIPFLAGS := $3 /" Makefile && make 2>&1 | grep -E "error" || true)
    ls -la "$d"/approximate-spmv-topk_amd/libtkspmv.so
else
    rounds=$2; rows=$3; cols=$4; nnz=$5; shift 5
    for r in $(seq 1 $rounds); do
        for v in "$@"; do
            d=_ab/$v; [ "$v" = . ] && d=.
            echo -n "$v: "; (cd $d && timeout -k 10 200 python tools/ablate_probe.py $rows $cols $nnz none 2>&1 | grep flags | head -1)
        done
    done
fi
