"""Time per query against K (1M x 1024, 20 nnz/row): one query per pass and the multi-query path."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
import torch
mod = _pkg.load()
m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(32)])
dxs = torch.from_numpy(xs).cuda()
for k in (1, 8, 20, 100, 200, 256, 300, 500, 1000, 1024):
    out = []
    for mq in (0, 8):
        eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=k, device=0, stream_replicas=3, multi_q=mq)
        f = eng.time_queries if mq == 0 else eng.time_multi
        f(dxs.data_ptr(), 32, 128)
        ns = min(f(dxs.data_ptr(), 32, 384) for _ in range(2))
        info = eng.info()
        out.append(f"{'1/pass' if mq == 0 else str(info['multi_q']) + '/pass'} {ns / 1e3:8.2f} us")
        eng.close()
    print(f"K={k:<5} groups {info['n_groups']:<5}" + "   ".join(out), flush=True)
