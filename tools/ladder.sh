#!/bin/bash
# Attribution ladder of the headline kernel: where do the microseconds of a query go?
#   tools/ladder.sh build            (CPU container) timing-only builds of the library under _ab/L1 .. _ab/L3 (-DTKSPMV_LADDER=n)
#   tools/ladder.sh run OUT [ROUNDS] (GPU box) every rung in turn, ROUNDS times, interleaved on this ONE box in its sustained state;
#                                    clocks / power / temperature sampled by a side process that never touches the GPU; then one
#                                    rocprofv3 --kernel-trace --stats pass over a run that launches nothing but headline queries.
# Rungs: L0 load-only (read_probe_kernel, same geometry: printed by every run as read_only_us); L1 + unpack / gather / multiply / scan /
# trigger (never taken); L2 + candidate path, thresholds, staging and ranking in LDS; L3 + records stored and drained; L4 + selections
# (the product with PACE=0); L5 the product with pauses by rank only (PACE_PERIOD=0); L6 the product (timetable or pauses by rank,
# whichever tkspmv_create measured faster).
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    for n in 1 2 3; do tools/ab_variants.sh build L$n "-DTKSPMV_LADDER=$n" & done
    wait
    exit 0
fi
out=${2:-gpurun_out/ladder}; rounds=${3:-3}
mkdir -p "$out"
python tools/clock_sampler.py "$out/clocks.csv" 0.05 &
sampler=$!
sleep 0.5
for r in $(seq 1 $rounds); do
    for n in 1 2 3; do
        TKSPMV_LIB=$PWD/_ab/L$n/approximate-spmv-topk_amd/libtkspmv.so timeout -k 10 240 python tools/ladder_probe.py L$n >> "$out/ladder.jsonl"
    done
    TKSPMV_PACE=0 timeout -k 10 240 python tools/ladder_probe.py L4_no_pacing >> "$out/ladder.jsonl"
    TKSPMV_PACE_PERIOD=0 timeout -k 10 240 python tools/ladder_probe.py L5_pauses_by_rank >> "$out/ladder.jsonl"
    timeout -k 10 240 python tools/ladder_probe.py L6_product >> "$out/ladder.jsonl"
    echo "round $r done: $(tail -1 "$out/ladder.jsonl" | cut -c1-200)"
done
kill $sampler 2>/dev/null || true
python tools/ladder_summary.py "$out" > "$out/summary.txt"
cat "$out/summary.txt"
