"""The read-only floor of the packet stream on this GPU (tkspmv_time_stream_read) against prefetch depth and per-packet
arithmetic (TKSPMV_READ_PROBE=depth,work), next to the batch kernel's time per query on the same box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

os.environ.setdefault("TKSPMV_F32_C12", "0")  # the depth / work variants of the probe are built for the 1536-byte (16-bit column word) packets
torch.cuda.init()
mod = _pkg.load()
m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
eng.enqueue_many(dxs.data_ptr(), 64, 256)
eng.synchronize()
t = sorted(eng.time_queries(dxs.data_ptr(), 64, 512) / 1e3 for _ in range(7))[3]
print(f"batch kernel: {t:.2f} us per query")
cases = ["8,0", "2,0", "3,0", "4,0", "6,0", "12,0", "16,0", "3,1", "3,32", "3,64", "3,96", "3,128", "4,64", "4,96", "6,64", "6,96", "8,1", "8,64", "8,96"]
for mp in ("1", "2", "0"):
    os.environ["TKSPMV_READ_PROBE_MAP"] = mp
    os.environ["TKSPMV_READ_PROBE"] = "8,0"
    r = sorted(eng.time_stream_read(64) / 1e3 for _ in range(5))[2]
    print(f"read probe, partition map {mp}: {r:6.2f} us per pass")
for c in (sys.argv[1:] or cases):
    os.environ["TKSPMV_READ_PROBE"] = c
    r = sorted(eng.time_stream_read(64) / 1e3 for _ in range(5))[2]
    print(f"read probe depth,work = {c:>6}: {r:6.2f} us per pass")
t = sorted(eng.time_queries(dxs.data_ptr(), 64, 512) / 1e3 for _ in range(7))[3]
print(f"batch kernel: {t:.2f} us per query")
eng.close()
