import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
import torch
mod = _pkg.load()
rows = int(os.environ.get("ROWS", "1000000"))
m = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(8)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=4)
print(eng.info())
for count in (1, 2, 3, 4, 8, 16):
    eng.enqueue_batch(dxs.data_ptr(), min(count, 8)); eng.synchronize()
    t0 = time.perf_counter()
    for rep in range(5):
        eng.enqueue_many(dxs.data_ptr(), 8, count)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"count {count}: {dt*1e6:.1f} us per call, {dt*1e6/count:.1f} us per query")
