#!/bin/bash
# Round-end measurement on the GPU box (run from the repository root through gpurun):
#   bash tools/profile_round.sh r01
# 1. plain bench line; 2. the same command under rocprofv3 --kernel-trace --stats; 3. separate --pmc passes for the
# memory-side counters (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains with --pmc);
# 4. the multi-query path alone (8 and 4 queries per pass) under --kernel-trace --stats; 5. single fused launches
# (tkspmv_run) under --kernel-trace --stats; 6. configs[4] (Q1.7 bytes, fp32 arithmetic) under --kernel-trace --stats;
# 7. one 125k-row shard through the sharded step; 8. configs[3] (10M rows) on one GPU, both under --kernel-trace --stats.
# 9. the shard rehearsal lines and the size sweep.
# Everything lands in gpurun_out/prof_<tag>/; tools/summarize_profile.py turns it into the files kept under profiles/.
set -u
TAG=${1:-r01}
PART=${2:-all}  # A: steps 1-3 (the bench line, its kernel trace, the counter passes); B: the rest; all: both (two calls fit gpurun's limit)
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
if [ "$PART" != B ]; then rm -rf "$OUT"; fi  # (a summary must never mix two runs)
mkdir -p "$OUT"
(while true; do sleep 60; echo "[alive $(date +%T)] $(ls -t "$OUT" | head -1)"; done) &
ALIVE=$!
trap "kill $ALIVE 2>/dev/null" EXIT
if [ "$PART" != B ]; then
python3 bench.py --steps 3000 --warmup 300 > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
# (the driver's own command line: ONE launch of 20 queries, start-up and tail included)
python3 bench.py --gpus 1 --steps 20 --warmup 5 --skip-warm --cpu-seconds 0 --traffic off > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"
cd /tmp && export TMPDIR=/tmp
# (--headline-only: the trace holds nothing but headline queries -- VERDICT r4 --, plus the 100 launches tkspmv_create measures its pacing
#  with, which tools/summarize_profile.py leaves out of <tag>_headline_launches.json)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace" -- python3 "$REPO/bench.py" --steps 2048 --warmup 256 --headline-only > "$OUT/bench_under_rocprofv3.json" 2> "$OUT/ktrace.err"
# (the counter passes run without tkspmv_create's measurement -- its launches would be counted --: with the period the plain run found)
PERIOD=$(python3 -c "import json,sys; print(json.loads([l for l in open('$OUT/bench_plain.json').read().splitlines() if l.startswith('{')][-1])['roofline'].get('pace_period_ns') or 0)")
i=0
for counters in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    TKSPMV_AUTOTUNE=0 TKSPMV_PACE=2 TKSPMV_PACE_LEVELS=6 TKSPMV_PACE_PERIOD=$PERIOD rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$OUT/pmc$i" -- python3 "$REPO/bench.py" --steps 320 --warmup 32 --headline-only > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err"
done
fi
if [ "$PART" = A ]; then cd "$REPO"; python3 tools/summarize_profile.py "$OUT" "$TAG"; exit 0; fi
cd /tmp && export TMPDIR=/tmp
# (one chain of launches, so that a launch's duration is the time of its pass; bench.py's figure overlaps two chains)
export TKSPMV_MULTI_CHAINS=1
for q in 8 4; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mtrace$q" -- python3 "$REPO/bench.py" --multi-only $q --steps 2048 --warmup 256 > "$OUT/multi$q.json" 2> "$OUT/mtrace$q.err"
done
unset TKSPMV_MULTI_CHAINS
# 5. the literal reference loop: single fused launches (tkspmv_run), one query at a time
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/single" -- python3 "$REPO/tools/single_probe.py" --plain 200 > "$OUT/single.log" 2> "$OUT/single.err"
# 6. BASELINE configs[4]: Q1.7 byte values with fp32 arithmetic, one query per pass over the row-per-lane byte stream (one chain)
export TKSPMV_MULTI_CHAINS=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/q17f" -- python3 "$REPO/tools/q17f_probe.py" --only > "$OUT/q17f.log" 2> "$OUT/q17f.err"
unset TKSPMV_MULTI_CHAINS
# 6b. the same in the mode bench.py reports it: two overlapping chains of launches (a launch's duration is then longer than the
#     time per query: two are in flight; the log carries the us per query of that very run)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/q17f2" -- python3 "$REPO/tools/q17f_probe.py" --only > "$OUT/q17f2.log" 2> "$OUT/q17f2.err"
# 7. one shard of a strong-scaled run (1/8 of the 1M-row matrix) through bench.py's sharded code path: the batch kernel with its
#    small-matrix settings (4 selector workgroups, workgroup-local thresholds, the repair launch behind every batch launch)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/shard125k" -- python3 "$REPO/bench.py" --total-rows 125000 --config3-rows 0 --steps 2048 --warmup 256 > "$OUT/shard125k.json" 2> "$OUT/shard125k.err"
# 8. BASELINE configs[3] on ONE GPU (10M rows) in the mode the bench reports it: back-to-back batch launches, two stream copies
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/config3" -- python3 "$REPO/bench.py" --rows 10000000 --replicas 2 --steps 128 --warmup 32 --cpu-seconds 0 --skip-warm --traffic off --reps 4 --multi-q > "$OUT/config3.json" 2> "$OUT/config3.err"
cd "$REPO"
# 9. the shards of a strong-scaled 1M-row run through bench.py's sharded code path on ONE GPU, and the size sweep
for rows in 125000 250000 500000; do
  timeout -k 10 300 python3 bench.py --total-rows $rows --config3-rows 0 --steps 2048 --warmup 256 > "$OUT/shard_rehearsal_$rows.json" 2> "$OUT/shard_rehearsal_$rows.err"
done
{ echo "# us per query of back-to-back queries (batch kernel), median of 5 timed batches of 256; tools/size_sweep.py"
  echo "## SWEEP=small (the shard sizes of a strong-scaled 1M-row matrix), defaults"; SWEEP=small timeout -k 10 300 python3 tools/size_sweep.py 2>/dev/null | grep F32
  echo "## the same with TKSPMV_SMALL_PACKETS=0 (one selector workgroup, partitions of 4+ packets, device-wide exchange)"; SWEEP=small TKSPMV_SMALL_PACKETS=0 timeout -k 10 300 python3 tools/size_sweep.py 2>/dev/null | grep F32
  echo "## SWEEP=mid, defaults"; SWEEP=mid timeout -k 10 300 python3 tools/size_sweep.py 2>/dev/null | grep F32
  echo "## SWEEP=mid, TKSPMV_LOCAL=0 (device-wide exchange on the same partition cut)"; SWEEP=mid TKSPMV_LOCAL=0 timeout -k 10 300 python3 tools/size_sweep.py 2>/dev/null | grep F32
  echo "## full sweep (sizes, k, value types), defaults"; timeout -k 10 600 python3 tools/size_sweep.py 2>/dev/null | grep -E "F32|F16|Q1_7"; } > "$OUT/size_sweep.txt"
python3 tools/summarize_profile.py "$OUT" "$TAG"
