#!/bin/bash
# Instruction-issue and LDS counters of the streaming kernels (run from the repository root through gpurun):
#   bash tools/pmc_issue.sh r01
# Separate rocprofv3 --pmc passes (few counters each; no trace domains besides --kernel-trace) over the headline path
# (batch kernel) and the multi-query path (8 and 4 queries per pass, one chain). tools/summarize_pmc_issue.py sums the
# counters per kernel into profiles/<tag>_pmc_issue.json.
set -u
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_issue_$TAG
rm -rf "$OUT"  # (a summary must never mix two runs)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export TKSPMV_MULTI_CHAINS=1
# (the counter passes run without tkspmv_create's measurement -- its launches would be counted --: the timetable's period of an
#  un-profiled run on this box first, handed to them as an option)
PERIOD=$(python3 "$REPO/bench.py" --steps 64 --warmup 32 --headline-only 2> "$OUT/period.err" | python3 -c "import json,sys; print(json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]).get('pace_period_ns') or 0)")
echo "timetable period of this box: $PERIOD ns" > "$OUT/period.txt"
i=0
for counters in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
                "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    TKSPMV_AUTOTUNE=0 TKSPMV_PACE=2 TKSPMV_PACE_LEVELS=6 TKSPMV_PACE_PERIOD=$PERIOD rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$OUT/batch$i" -- python3 "$REPO/bench.py" --steps 320 --warmup 32 --headline-only > "$OUT/batch$i.json" 2> "$OUT/batch$i.err"
    for q in 8 4; do
        rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$OUT/multi${q}_$i" -- python3 "$REPO/bench.py" --multi-only $q --steps 320 --warmup 32 > "$OUT/multi${q}_$i.json" 2> "$OUT/multi${q}_$i.err"
    done
done
cd "$REPO"
python3 tools/summarize_pmc_issue.py "$OUT" "$TAG"  # -> profiles/ and $OUT/summary/ (copy the latter when run through gpurun)
