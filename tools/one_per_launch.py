"""One query per launch through the batch kernel (tkspmv_enqueue_batch with count = 1, back to back) against the fused single-query
launch (tkspmv_enqueue): device time per query by a hipEvent pair around 400 launches. Development probe."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

mod = _pkg.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(16)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=4)
out_i = torch.zeros(16, 100, dtype=torch.int32, device="cuda")
out_v = torch.zeros(16, 100, dtype=torch.float32, device="cuda")
s = torch.cuda.Stream()
for name in ("batch(1)", "fused"):
    res = []
    for rep in range(5):
        with torch.cuda.stream(s):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(40):  # warm
                if name == "fused":
                    eng.enqueue(dxs[i % 16].data_ptr(), out_i[i % 16].data_ptr(), out_v[i % 16].data_ptr(), s.cuda_stream)
                else:
                    eng.enqueue_batch(dxs[i % 16].data_ptr(), 1, out_i[i % 16].data_ptr(), out_v[i % 16].data_ptr(), s.cuda_stream)
            e0.record(s)
            for i in range(400):
                if name == "fused":
                    eng.enqueue(dxs[i % 16].data_ptr(), out_i[i % 16].data_ptr(), out_v[i % 16].data_ptr(), s.cuda_stream)
                else:
                    eng.enqueue_batch(dxs[i % 16].data_ptr(), 1, out_i[i % 16].data_ptr(), out_v[i % 16].data_ptr(), s.cuda_stream)
            e1.record(s)
        s.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 400)
    print(f"{rows} rows, {name}: {sorted(res)[2]:.2f} us per query (median of 5 x 400 launches back to back)")
eng.close()
