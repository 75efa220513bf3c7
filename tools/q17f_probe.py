"""BASELINE configs[4] (1M x 512, 40 nnz/row, K=100; ROWS/COLS/NNZ override) with Q1.7 values and fp32 arithmetic
(TKSPMV_Q1_7_F32): time per query of the batch kernel over the wave-BSCSR byte stream and of the row-per-lane kernels over
byte chunks (1, 4, 8 queries per pass), next to fp32 values; precision@100 against the fp32 gold (3 queries).
TKSPMV_MULTI_CHAINS=1: one chain of launches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg
import torch
mod = _pkg.load()
rows, cols, nnz = (int(os.environ.get(k, d)) for k, d in (("ROWS", 1000000), ("COLS", 512), ("NNZ", 40)))
m = mod.generate_matrix(rows, cols, nnz, "gamma", 5)
xs = np.stack([mod.create_sample_vector(cols, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
ONLY = "--only" in sys.argv  # profiler runs: nothing but configs[4]'s measured path (Q1.7 bytes, one query per pass)
for name, mq in ((("Q1_7_F32", 1),) if ONLY else (("F32", 0), ("Q1_7_F32", 0), ("F32", 1), ("Q1_7_F32", 1), ("Q1_7_F32", 4), ("Q1_7_F32", 8))):
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, precision=getattr(mod, name), stream_replicas=4, multi_q=mq)
    info = eng.info()
    f = eng.time_queries if mq == 0 else eng.time_multi
    f(dxs.data_ptr(), 64, 640)
    ns = min(f(dxs.data_ptr(), 64, 1920) for _ in range(3))
    stream = info["multi_bytes"] if mq else info["packed_bytes"]
    print(f"{name:9s} {'batch kernel' if mq == 0 else 'row per lane, %d per pass' % mq:24s} stream {stream/1e6:7.1f} MB  algorithmic "
          f"{info['algorithmic_bytes']/1e6:6.1f} MB  {ns/1e3:6.2f} us/query  {info['algorithmic_bytes']/ns:6.0f} GB/s algorithmic = "
          f"{info['algorithmic_bytes']/ns/80:4.1f} % of 8 TB/s", flush=True)
    eng.close()
if os.environ.get("PRECISION", "1") == "1" and not ONLY:
    import oracle_lib as O
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, precision=mod.Q1_7_F32)
    p = []
    for q in range(3):
        eng.reset(xs[q]); eng(); val, idx = eng.read_result()
        gi, gv = O.gold_topk(m.row, m.col, m.val, xs[q], 100)
        p.append(len(set(idx.tolist()) & set(gi.tolist())) / 100)
    print("precision@100 vs the fp32 gold:", p)
    eng.close()
