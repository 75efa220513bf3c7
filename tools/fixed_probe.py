"""The reference's fixed-point builds (FIXED_WIDTH 20/21/25/26/32, test_spmv_topk.py:42-47) on BASELINE configs[1]'s
matrix (1M x 1024, 20 nnz/row, K=100; ROWS/COLS/NNZ override the shape): time per query of the batch kernel and
precision@100 / Kendall's tau / NDCG against the fp32 gold (the reference's accuracy metrics, plot_errors.py:182-231),
averaged over N_QUERIES query vectors."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg
import torch
import oracle_lib as O
mod = _pkg.load()
from importlib import import_module
ex = import_module("approximate_spmv_topk_amd.experiments")
rows, cols, nnz, nq = (int(os.environ.get(k, d)) for k, d in (("ROWS", 1000000), ("COLS", 1024), ("NNZ", 20), ("N_QUERIES", 8)))
m = mod.generate_matrix(rows, cols, nnz, "gamma", 2)
xs = np.stack([mod.create_sample_vector(cols, True, False, True, 1000 + i) for i in range(nq)])
dxs = torch.from_numpy(xs).cuda()
gold = [O.gold_topk(m.row, m.col, m.val, xs[q], 100) for q in range(nq)]
for name, width in (("F32", 0), ("F16", 0), ("FIXED", 32), ("FIXED", 26), ("FIXED", 25), ("FIXED", 21), ("FIXED", 20), ("FIXED", 16)):
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, precision=getattr(mod, name), fixed_width=width,
                   stream_replicas=4)
    info = eng.info()
    ns = eng.time_queries(dxs.data_ptr(), nq, 640)
    ns = eng.time_queries(dxs.data_ptr(), nq, 640)
    prec, tau, ndcg = [], [], []
    for q in range(nq):
        eng.reset_device(dxs[q].data_ptr())
        eng()
        val, idx = eng.read_result()
        gi, gv = gold[q]
        prec.append(ex.precision_at(gi.tolist(), idx.tolist(), 100))
        tau.append(ex.kendall_tau(gi.tolist(), idx.tolist()))
        ndcg.append(ex.ndcg(gi.tolist(), gv.tolist(), idx.tolist(), val.tolist())[0])
    label = name + (str(width) if width else "")
    print(f"{label:8s} {ns/1e3:6.2f} us/query  {info['algorithmic_bytes']/ns:6.0f} GB/s algorithmic  precision@100 {np.mean(prec):.4f}  "
          f"kendall-tau {np.mean(tau):.4f}  ndcg {np.mean(ndcg):.5f}")
    eng.close()
