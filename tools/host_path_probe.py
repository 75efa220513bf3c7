"""Cost of each call of the reference loop's host boundary (set_query, run, read), medians over 300 iterations."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
import torch
mod = _pkg.load()
m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(8)])
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=4, impl=int(os.environ.get("IMPL", "0")))
ts = {"reset": [], "run": [], "read": [], "kernel_us": []}
for i in range(320):
    t0 = time.perf_counter(); eng.reset(xs[i % 8]); t1 = time.perf_counter(); ns = eng(); t2 = time.perf_counter(); eng.read_result(); t3 = time.perf_counter()
    ts["reset"].append((t1 - t0) * 1e6); ts["run"].append((t2 - t1) * 1e6); ts["read"].append((t3 - t2) * 1e6); ts["kernel_us"].append(ns / 1e3)
print("IMPL =", os.environ.get("IMPL", "0"), "TKSPMV_HOST_PATH =", os.environ.get("TKSPMV_HOST_PATH", "(default 1)"), "TKSPMV_RUN_EVENTS =", os.environ.get("TKSPMV_RUN_EVENTS", "(default)"))
for k, v in ts.items():
    v = np.array(v[20:])
    print(f"  {k:10s} p50 {np.percentile(v, 50):8.1f}  p95 {np.percentile(v, 95):8.1f}  min {v.min():8.1f}")
eng.close()
