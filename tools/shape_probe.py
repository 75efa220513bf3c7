"""One-query-per-pass batch kernel against the multi-query path (4 and 8 queries per pass) over a few shapes and K:
time per query, cache-defeated rotation. Looks for pathologies (small matrices, short partitions, large K)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
import torch
mod = _pkg.load()
shapes = [(10000, 1024, 20, "gamma", 100), (100000, 1024, 20, "gamma", 100), (1000000, 1024, 20, "gamma", 8),
          (1000000, 512, 40, "gamma", 100), (1000000, 1024, 20, "uniform", 100), (3000000, 1024, 20, "gamma", 100),
          (1000000, 1024, 20, "gamma", 500), (2000000, 300, 25, "gamma", 100)]
for rows, cols, nnz, dist, k in shapes:
    m = mod.generate_matrix(rows, cols, nnz, dist, 2)
    xs = np.stack([mod.create_sample_vector(cols, True, False, True, 1000 + i) for i in range(32)])
    dxs = torch.from_numpy(xs).cuda()
    out = []
    for mq in (0, 4, 8):
        eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=k, device=0, stream_replicas=3, multi_q=mq)
        f = eng.time_queries if mq == 0 else eng.time_multi
        f(dxs.data_ptr(), 32, 320)
        ns = min(f(dxs.data_ptr(), 32, 960) for _ in range(2))
        out.append(f"{'1/pass' if mq == 0 else str(mq) + '/pass'} {ns / 1e3:7.2f} us" + ("" if mq == 0 or eng.info()["multi_q"] else " (fallback)"))
        eng.close()
    print(f"{rows:>8} x {cols:<5} nnz {nnz:<3} {dist:<8} K={k:<4} " + "   ".join(out), flush=True)
