# One-GPU rehearsal of the shards of a strong-scaled 1M-row run (bench.py's sharded code path with ONE shard of 1/8, 1/4, 1/2
# of the matrix: local batch kernel + the native exchange with a one-rank communicator), and the size sweep (run on the GPU box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rows in 125000 250000 500000; do
  timeout -k 10 300 python bench.py --total-rows $rows --config3-rows 0 --steps 2048 --warmup 256 > gpurun_out/r03_shard_rehearsal_$rows.json 2> gpurun_out/r03_shard_rehearsal_$rows.err || exit 1
done
{ echo "# us per query of back-to-back queries (batch kernel), median of 5 timed batches of 256; tools/size_sweep.py"; echo "## SWEEP=small (the shard sizes of a strong-scaled 1M-row matrix), round-3 defaults"; SWEEP=small timeout -k 10 300 python tools/size_sweep.py 2>/dev/null | grep F32;
  echo "## the same with TKSPMV_SMALL_PACKETS=0 (round 2's behaviour: one selector workgroup, partitions of 4+ packets, device-wide exchange)"; SWEEP=small TKSPMV_SMALL_PACKETS=0 timeout -k 10 300 python tools/size_sweep.py 2>/dev/null | grep F32;
  echo "## SWEEP=mid, round-3 defaults"; SWEEP=mid timeout -k 10 300 python tools/size_sweep.py 2>/dev/null | grep F32;
  echo "## SWEEP=mid, TKSPMV_SMALL_PACKETS=0"; SWEEP=mid TKSPMV_SMALL_PACKETS=0 timeout -k 10 300 python tools/size_sweep.py 2>/dev/null | grep F32;
  echo "## full sweep (sizes, k, value types), round-3 defaults"; timeout -k 10 600 python tools/size_sweep.py 2>/dev/null | grep -E "F32|F16|Q1_7"; } > gpurun_out/r03_size_sweep.txt
tail -40 gpurun_out/r03_size_sweep.txt
