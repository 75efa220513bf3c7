"""What the host adds around a 20-query timed region (bench.py --steps 20): wall clock of each call against the device time."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

torch.cuda.init()
mod = _pkg.load()
m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
eng.enqueue_many(dxs.data_ptr(), 64, 5)
eng.synchronize()
torch.cuda.synchronize()
pc = time.perf_counter
rows = []
for rep in range(12):
    t0 = pc(); ns = eng.time_queries(dxs.data_ptr(), 64, 20); t1 = pc(); eng.synchronize(); t2 = pc(); torch.cuda.synchronize(); t3 = pc()
    rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, ns * 20 / 1e3))
rows = np.array(rows[2:])
print("time_queries(20): wall %.1f us (device %.1f us), then eng.synchronize %.1f us, torch.cuda.synchronize %.1f us" % tuple(np.median(rows, axis=0)[[0, 3, 1, 2]]))
rows = []
for rep in range(12):
    t0 = pc(); eng.enqueue_many(dxs.data_ptr(), 64, 20); t1 = pc(); eng.synchronize(); t2 = pc(); torch.cuda.synchronize(); t3 = pc()
    rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
rows = np.array(rows[2:])
print("enqueue_many(20): wall %.1f us, then eng.synchronize %.1f us, torch.cuda.synchronize %.1f us; total %.1f" % (tuple(np.median(rows, axis=0)) + (np.median(rows.sum(axis=1)),)))
eng.close()
