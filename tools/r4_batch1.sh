#!/bin/bash
# Round 4: the reworked batch kernel -- parity first (the tests that exercise its modes), then A/B against round 3 on this box.
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r4m
rm -rf "$OUT"; mkdir -p "$OUT"
(while true; do sleep 60; echo "[alive $(date +%T)] $(tail -c 200 "$OUT/tests.log" 2>/dev/null | tr -d "\n" | tail -c 80)"; done) &
ALIVE=$!
trap "kill $ALIVE 2>/dev/null" EXIT
timeout -k 10 1000 python3 -m pytest --timeout=120 --timeout-method=thread tests -x -q -m gpu -k "not ten_million and not test_round3_batch_modes" > "$OUT/tests.log" 2>&1
rc=$?
tail -6 "$OUT/tests.log"
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" "$OUT/tests.log" | head -20; exit $rc; }
bash tools/ab_variants.sh run 2 1000000 1024 20 r3 . > "$OUT/ab_1m.log" 2>&1; cat "$OUT/ab_1m.log"
bash tools/ab_variants.sh run 1 125000 1024 20 r3 . > "$OUT/ab_125k.log" 2>&1; cat "$OUT/ab_125k.log"
