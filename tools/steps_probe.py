"""How a short timed region (the driver runs bench.py --steps 20 --warmup 5) splits into per-launch overhead and
per-query time: time_queries(n) for several n after (a) a 5-query warm-up right after create, (b) a long warm-up.
Prints median / min over repeats and a least-squares fit a + b*n."""
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
if os.environ.get("XS") == "uniform":
    xs = np.random.RandomState(11).rand(64, 1024).astype(np.float32)
else:  # bench.py's queries
    xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
eng.enqueue_many(dxs.data_ptr(), 64, 5)
eng.synchronize()
first = eng.time_queries(dxs.data_ptr(), 64, 20) / 1e3
print(f"cold (create, 5 warm-up queries, then 20 timed): {first:.2f} us/query")
for rep in range(3):
    print(f"  again: {eng.time_queries(dxs.data_ptr(), 64, 20) / 1e3:.2f}")
eng.enqueue_many(dxs.data_ptr(), 64, 2048)
eng.synchronize()
ns, med, mn = [], [], []
for n in (1, 2, 4, 8, 12, 16, 20, 24, 32, 64, 128):
    t = [eng.time_queries(dxs.data_ptr(), 64, n) * n / 1e3 for _ in range(20)]
    ns.append(n); med.append(np.median(t)); mn.append(np.min(t))
    print(f"n={n:4d}: region median {np.median(t):8.1f} us  min {np.min(t):8.1f}  per query {np.median(t) / n:6.2f}")
sel = [i for i, n in enumerate(ns) if n <= 32]
b, a = np.polyfit(np.array(ns)[sel], np.array(med)[sel], 1)
print(f"fit over n <= 32: region = {a:.1f} us + {b:.2f} us * n")
# after an idle gap (the host does something else for 50 ms)
for gap in (0.0, 0.01, 0.1, 1.0):
    t = []
    for _ in range(5):
        time.sleep(gap)
        t.append(eng.time_queries(dxs.data_ptr(), 64, 20) / 1e3)
    print(f"idle {gap * 1e3:6.0f} ms before a 20-query region: median {np.median(t):.2f} us/query (min {np.min(t):.2f}, max {np.max(t):.2f})")
eng.close()
