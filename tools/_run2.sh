mkdir -p gpurun_out/multi
(while true; do sleep 60; echo "[alive] $(tail -c 100 gpurun_out/multi/tests.log | tr -d '\n')"; done) &
A=$!
timeout -k 10 1100 python3 -m pytest --timeout=300 --timeout-method=thread tests -x -q -m gpu -k "multi or row_per_lane or q17 or one_query_per_pass or device_pack or dist_batch or sharded" > gpurun_out/multi/tests.log 2>&1
rc=$?
kill $A
tail -8 gpurun_out/multi/tests.log
exit $rc
