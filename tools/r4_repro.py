import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
mod = _pkg.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=4)
print("info", eng.info()["batch_mode"] & 0xFFFF, flush=True)
for n in (5, 20, 32, 33, 64):
    t = eng.time_queries(dxs.data_ptr(), 64, n)
    print("time_queries", n, t / 1e3, eng.debug_counters(), flush=True)
v, i = eng.read_result()
print("read ok", i[:4], flush=True)
print("read probe", eng.time_stream_read(4) / 1e3, flush=True)
eng.close()
print("done", flush=True)
