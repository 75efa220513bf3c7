#!/bin/bash
set -u
cat > /tmp/exact2.py <<'PY'
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import _pkg
mod = _pkg.load()
m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
eng.enqueue_many(dxs.data_ptr(), 64, 512); eng.synchronize()
t = sorted(eng.time_queries(dxs.data_ptr(), 64, 512) / 1e3 for _ in range(5))
print(f"{os.path.basename(os.getcwd())} STATS={os.environ.get('TKSPMV_STATS','-')}: median {t[2]:.2f} us/query", flush=True)
eng.profile(dxs.data_ptr(), 64, 4)
eng.close()
PY
for d in . _ab/r3; do (cd $d && TKSPMV_LOCAL=0 TKSPMV_STATS=1 timeout -k 10 120 python3 /tmp/exact2.py 2>&1 | grep -E "median|stats") || exit 1; done
