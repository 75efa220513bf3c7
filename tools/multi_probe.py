"""Several queries per pass over the matrix (tkspmv_enqueue_multi) on BASELINE configs[1] (1M x 1024, 20 nnz/row, K=100,
cache-defeated rotation): time per query for 1, 2, 4, 8 queries per pass (MQ="0 1 2 4 8"; 0 = the one-query-per-pass
batch kernel). TKSPMV_STATS=1 adds the candidate-path counters, TKSPMV_MULTI_CHAINS=1 runs one chain of launches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
import torch
mod = _pkg.load()
rows, cols, nnz, k = (int(os.environ.get(n, d)) for n, d in (("ROWS", 1000000), ("COLS", 1024), ("NNZ", 20), ("K", 100)))
m = mod.generate_matrix(rows, cols, nnz, "gamma", 2)
xs = np.stack([mod.create_sample_vector(cols, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
for mq in os.environ.get("MQ", "0 1 2 4 8").split():
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=k, device=0, stream_replicas=4, multi_q=int(mq))
    info = eng.info()
    f = eng.time_queries if mq == "0" else eng.time_multi
    f(dxs.data_ptr(), 64, 640)
    ns = min(f(dxs.data_ptr(), 64, 1920) for _ in range(3))
    print(f"{'batch kernel (1 query per pass)' if mq == '0' else 'multi_q = ' + mq:32s} {ns/1e3:6.2f} us/query  {1e9/ns:8.0f} queries/s  "
          f"{info['algorithmic_bytes']/ns:6.0f} GB/s per-query algorithmic", flush=True)
    eng.close()
