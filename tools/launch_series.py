"""Launch by launch: the time per query of N consecutive launches of 32 queries (one hipEvent between launches, all enqueued before the
first wait), in order -- does the kernel's time alternate or drift from launch to launch?   python tools/launch_series.py [N]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
torch.cuda.init()
mod = _pkg.load()
m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=int(os.environ.get("REPLICAS", 4)))
nql = int(os.environ.get("QUERIES_PER_LAUNCH", 32))
eng.time_query_batches(dxs.data_ptr(), 64, nql, 16)
v = [x / 1e3 for x in eng.time_query_batches(dxs.data_ptr(), 64, nql, n)]
c = eng.debug_counters()
print(f"pace {c['pace_quantum']}x{c['pace_levels']} period {c.get('pace_period_ns')} replicas {os.environ.get('REPLICAS', 4)} carry {os.environ.get('TKSPMV_PACE_CARRY', '1')}: median {np.median(v):.2f} p95 {np.percentile(v, 95):.2f} (x{np.percentile(v, 95) / np.median(v):.3f}) min {min(v):.2f} max {max(v):.2f} mean {np.mean(v):.2f}")
print(" ".join(f"{x:.1f}" for x in v))
w = [x / 1e3 for x in eng.time_query_batches(dxs.data_ptr(), 64, 256, 12)]
print("reps of 256 queries:", " ".join(f"{x:.2f}" for x in w))
eng.close()
