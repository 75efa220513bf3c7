"""us per query of back-to-back queries (batch kernel) over matrix sizes and k: for A/B runs of two builds on one box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

torch.cuda.init()
mod = _pkg.load()
out = []
SMALL = [(r, 1024, 20, 100, "F32") for r in (50000, 125000, 250000, 500000)]  # SWEEP=small: the shard sizes of a strong-scaled 1M-row matrix
MID = [(r, 1024, 20, 100, "F32") for r in (400000, 600000, 800000, 1000000)]  # SWEEP=mid: where the small-matrix settings stop paying
WIDE = [(r, 1024, 20, 100, "F32") for r in (250000, 500000, 750000, 1000000, 1500000, 2000000, 3000000)]  # SWEEP=wide: pacing by size
if os.environ.get("SWEEP_ROWS"):
    WIDE = [(int(r), int(os.environ.get("SWEEP_COLS", "1024")), int(os.environ.get("SWEEP_NNZ", "20")), int(os.environ.get("SWEEP_K", "100")),
             os.environ.get("SWEEP_PREC", "F32")) for r in os.environ["SWEEP_ROWS"].split(",")]
for rows, cols, nnz, k, prec in SMALL if os.environ.get("SWEEP") == "small" else MID if os.environ.get("SWEEP") == "mid" else WIDE if os.environ.get("SWEEP") == "wide" else [(50000, 1024, 20, 100, "F32"), (200000, 1024, 20, 100, "F32"), (1000000, 1024, 20, 100, "F32"),
                                 (1000000, 1024, 20, 10, "F32"), (1000000, 1024, 20, 200, "F32"), (1000000, 512, 40, 100, "F32"),
                                 (1000000, 512, 40, 100, "Q1_7"), (1000000, 1024, 20, 100, "F16"), (2000000, 1024, 20, 100, "F32"),
                                 (3000000, 1024, 20, 100, "F32"), (1000000, 4096, 20, 100, "F32")]:
    m = mod.generate_matrix(rows, cols, nnz, "gamma", 2)
    xs = np.stack([mod.create_sample_vector(cols, True, False, True, 1000 + i) for i in range(16)])
    dxs = torch.from_numpy(xs).cuda()
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=k, device=0, stream_replicas=4, precision=getattr(mod, prec))
    eng.enqueue_many(dxs.data_ptr(), 16, 128)
    eng.synchronize()
    t = sorted(eng.time_queries(dxs.data_ptr(), 16, 256) / 1e3 for _ in range(5))[2]
    out.append(f"{rows}x{cols} nnz/row {nnz} k={k} {prec}: {t:.2f}")
    eng.close()
print("\n".join(out))
