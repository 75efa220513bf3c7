"""How a gap of host-side idleness changes the next batch launches (32 queries each, timed one by one): after a warm burst, sleep T,
then three launches. Development probe behind DESIGN.md 3.0's note on timing.kernel_us_p95.  python tools/gap_probe.py [ROWS]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

mod = _pkg.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=4)
for gap_ms in (0.0, 0.05, 0.2, 1.0, 5.0, 20.0, 100.0):
    res = []
    for rep in range(7):
        eng.time_queries(dxs.data_ptr(), 64, 640)  # warm burst: 20 launches
        if gap_ms:
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < gap_ms:
                pass
        res.append([eng.time_queries(dxs.data_ptr(), 64, 32) / 1e3 for _ in range(4)])
    r = np.median(np.array(res), axis=0)
    print(f"gap {gap_ms:6.2f} ms: the next four launches take {r[0]:.2f} {r[1]:.2f} {r[2]:.2f} {r[3]:.2f} us per query (medians of 7)")
eng.close()
