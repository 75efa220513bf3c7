#!/bin/bash
# exact mode (device-wide exchange) at 1M rows: selector workgroups 1 / 4, this build against round 3's, same box
# (needs the other tree next to this one: git worktree add -f _ab/r3 <round 3's last commit> && make -C _ab/r3; _ab/ is git-ignored
#  and travels to the GPU box with the snapshot; git worktree remove --force _ab/r3 afterwards)
set -u
OUT=gpurun_out/exact; mkdir -p $OUT
cat > /tmp/exact.py <<'PY'
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import _pkg
mod = _pkg.load()
rows = int(sys.argv[1])
m = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
eng.enqueue_many(dxs.data_ptr(), 64, 1024); eng.synchronize()
t = sorted(eng.time_queries(dxs.data_ptr(), 64, 512) / 1e3 for _ in range(7))
i = eng.info()
print(f"{os.path.basename(os.getcwd()):8s} rows {rows} LOCAL={os.environ.get('TKSPMV_LOCAL','-')} SELECTORS={os.environ.get('TKSPMV_SELECTORS','-')} SMALL={os.environ.get('TKSPMV_SMALL_PACKETS','-')}: median {t[3]:.2f} us/query (min {t[0]:.2f})  mode {i['batch_mode'] & 0xFFFF:#x} parts {i['n_wave_partitions']}", flush=True)
eng.close()
PY
for rows in 1000000; do
for sel in 4 1; do
  for d in . _ab/r3; do
    (cd $d && TKSPMV_LOCAL=0 TKSPMV_SELECTORS=$sel timeout -k 10 120 python3 /tmp/exact.py $rows 2>&1 | grep rows) || exit 1
  done
done
for d in . _ab/r3; do (cd $d && TKSPMV_SMALL_PACKETS=0 timeout -k 10 120 python3 /tmp/exact.py $rows 2>&1 | grep rows) || exit 1; done
for d in . _ab/r3; do (cd $d && timeout -k 10 120 python3 /tmp/exact.py $rows 2>&1 | grep rows) || exit 1; done
done
for d in . _ab/r3; do (cd $d && timeout -k 10 120 python3 /tmp/exact.py 3000000 2>&1 | grep rows) || exit 1; done
