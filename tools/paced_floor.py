"""The load-only probe on a timetable (option READ_PROBE_PERIOD): time per pass against the period, next to the unpaced probe and
the batch kernel -- is the unpaced probe a floor on this box?   python tools/paced_floor.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

torch.cuda.init()
mod = _pkg.load()
rows = int(os.environ.get("ROWS", 1000000))
m = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
eng.time_query_batches(dxs.data_ptr(), 64, 256, 8)
reps = [v / 1e3 for v in eng.time_query_batches(dxs.data_ptr(), 64, 256, 12)][2:]
print(f"batch kernel, sustained median: {np.median(reps):.2f} us per query; period in force {eng.debug_counters()['pace_period_ns']} ns")
mod.set_option("READ_PROBE_PERIOD", None)
f0 = sorted(eng.time_stream_read(64) / 1e3 for _ in range(5))[2]
print(f"load-only probe, unpaced: {f0:.2f} us per pass")
best = (f0, 0)
for rel in (0.88, 0.90, 0.92, 0.94, 0.96, 0.98, 1.0, 1.02):
    ns = int(f0 * 1000 * rel)
    mod.set_option("READ_PROBE_PERIOD", str(ns))
    t = sorted(eng.time_stream_read(64) / 1e3 for _ in range(5))[2]
    best = min(best, (t, ns))
    print(f"load-only probe, period {ns} ns ({rel:.2f} x unpaced): {t:.2f} us per pass")
mod.set_option("READ_PROBE_PERIOD", None)
print(f"paced floor: {best[0]:.2f} us per pass at a period of {best[1]} ns")
eng.close()
