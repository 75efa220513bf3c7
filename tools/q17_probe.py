"""Reduced-precision value streams on BASELINE configs[4] (1M x 512, 40 nnz/row, K=100; ROWS/COLS/NNZ override the
shape): time per query of the batch kernel for fp32, fp16 and Q1.7 values."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
import torch
mod = _pkg.load()
rows, cols, nnz = (int(os.environ.get(k, d)) for k, d in (("ROWS", 1000000), ("COLS", 512), ("NNZ", 40)))
m = mod.generate_matrix(rows, cols, nnz, "gamma", 3)
xs = np.stack([mod.create_sample_vector(cols, True, False, True, 1000 + i) for i in range(8)])
dxs = torch.from_numpy(xs).cuda()
for name in ("F32", "F16", "Q1_7", "Q1_7_WIDE"):
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, precision=getattr(mod, name), stream_replicas=4)
    info = eng.info()
    ns = eng.time_queries(dxs.data_ptr(), 8, 640)
    ns = eng.time_queries(dxs.data_ptr(), 8, 640)
    print(f"{name:10s} packed {info['packed_bytes']/1e6:7.1f} MB  algorithmic {info['algorithmic_bytes']/1e6:7.1f} MB  {ns/1e3:6.2f} us/query  "
          f"{info['algorithmic_bytes']/ns:7.0f} GB/s algorithmic")
    eng.close()
