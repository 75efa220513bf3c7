#!/bin/bash
# The whole GPU suite with a per-test timeout (a hung kernel names its test instead of eating the box's time), a heartbeat
# for the runner's silence watchdog, and a non-zero exit if the runtime reported a memory fault.
set -u
OUT=gpurun_out/${1:-gputests}; mkdir -p $OUT
(while true; do sleep 60; echo "[alive $(date +%T)] $(tail -c 120 "$OUT/tests.log" 2>/dev/null | tr -d "\n")"; done) &
ALIVE=$!
trap "kill $ALIVE 2>/dev/null" EXIT
timeout -k 10 1100 python3 -m pytest --timeout=300 --timeout-method=thread tests -x -q -m gpu ${2:-} > "$OUT/tests.log" 2>&1
rc=$?
tail -8 "$OUT/tests.log"
grep -q "Memory access fault" "$OUT/tests.log" && exit 9
exit $rc
