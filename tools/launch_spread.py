"""Spread of single batch launches (32 queries each, timed one by one) and how many threshold checks failed meanwhile
(tkspmv_debug_counters): what stands behind bench.py's `timing.kernel_us_p95` on the driver's command line. Development probe.
  python tools/launch_spread.py [ROWS] [LAUNCHES]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

mod = _pkg.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
m = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=4)
eng.time_queries(dxs.data_ptr(), 64, 256)
c0 = eng.debug_counters()
t = np.array([eng.time_queries(dxs.data_ptr(), 64, 32) / 1e3 for _ in range(n)])
c1 = eng.debug_counters()
q = lambda p: float(np.percentile(t, p))
print(f"{rows} rows, {n} launches of 32 queries, us per query: min {t.min():.2f} p10 {q(10):.2f} p50 {q(50):.2f} p90 {q(90):.2f} p95 {q(95):.2f} "
      f"p99 {q(99):.2f} max {t.max():.2f}; p95/p50 {q(95) / q(50):.3f}; checks failed meanwhile: {c1['checks_failed'] - c0['checks_failed']} "
      f"of {32 * n} selections (suspension now {c1['suspension_length']}, {c1['suspended_for']} to go); mode {eng.info()['batch_mode'] & 0xFFFF:#x}")
slow = np.argsort(t)[-8:]
print("slowest launches (index: us):", ", ".join(f"{i}: {t[i]:.1f}" for i in sorted(slow)))
eng.close()
