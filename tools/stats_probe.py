"""Candidate-path statistics of the batch kernel on BASELINE configs[1] (TKSPMV_STATS=1: the DBG instantiation):
packets that took the candidate path and rows offered per query, candidates reaching the selection."""
import os
import sys

os.environ["TKSPMV_STATS"] = "1"
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
eng.enqueue_many(dxs.data_ptr(), 64, 256)
eng.synchronize()
p = eng.profile(dxs.data_ptr(), 64, 256)
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in p.items()})
print("packets per query:", eng.info()["n_packets"])
eng.close()
