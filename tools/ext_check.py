"""tkspmv_time_queries: the event pair on the region's first and last kernel (EXT_EVENTS=1, hipExtLaunchKernelGGL) against the pair
recorded around the launches (EXT_EVENTS=0), alternating, for regions of 1..4 launches: 3-6 us apart whatever the length."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
torch.cuda.init()
mod = _pkg.load()
m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
eng.time_query_batches(dxs.data_ptr(), 64, 256, 16)
for n in (20, 32, 64, 128):
    res = {"1": [], "0": []}
    for i in range(12):
        for ext in ("1", "0"):
            os.environ["TKSPMV_EXT_EVENTS"] = ext
            res[ext].append(eng.time_queries(dxs.data_ptr(), 64, n) * n / 1e3)
    print(n, "queries: region us, ext events median %.1f min %.1f | recorded pair median %.1f min %.1f" % (np.median(res["1"]), min(res["1"]), np.median(res["0"]), min(res["0"])))
eng.close()
