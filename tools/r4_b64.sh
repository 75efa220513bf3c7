#!/bin/bash
# launches of up to 64 queries against launches of up to 32 (TKSPMV_BATCH_MAX), same build, same box; then the new tests
set -u
OUT=gpurun_out/b64; mkdir -p $OUT
cat > /tmp/b64.py <<'PY'
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import _pkg
mod = _pkg.load()
m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
engs = {}
for bm in ("32", "64"):
    mod.set_option("BATCH_MAX", bm)
    engs[bm] = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
    engs[bm].enqueue_many(dxs.data_ptr(), 64, 1024); engs[bm].synchronize()
for rnd in range(3):
    for bm, eng in engs.items():
        t = np.asarray(eng.time_query_batches(dxs.data_ptr(), 64, 256, 24)) / 1e3
        print(f"BATCH_MAX={bm}: sustained 24 x 256 queries: median {np.median(t[2:]):.2f} us/query  p95 {np.percentile(t[2:], 95):.2f}  min {t.min():.2f}", flush=True)
for bm, eng in engs.items():
    for n in (20, 64, 128):
        t = [eng.time_queries(dxs.data_ptr(), 64, n) / 1e3 for _ in range(15)]
        print(f"BATCH_MAX={bm}: region of {n}: median {np.median(t):.2f} us/query", flush=True)
    print(bm, eng.debug_counters())
    eng.close()
PY
timeout -k 10 300 python3 /tmp/b64.py > $OUT/ab.log 2>&1 || { tail -5 $OUT/ab.log; exit 1; }
cat $OUT/ab.log
timeout -k 10 900 python3 -m pytest --timeout=300 -x -q -m gpu tests/test_gpu_single.py tests/test_gpu_engine.py tests/test_gpu_local_fuzz.py tests/test_gpu_resident.py tests/test_gpu_configs.py > $OUT/tests.log 2>&1; rc=$?
tail -5 $OUT/tests.log
grep -q "Memory access fault" $OUT/tests.log && exit 9
exit $rc
