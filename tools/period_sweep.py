"""Pacing by the clock (option PACE_PERIOD): time per query against the timetable's period, the period given RELATIVE to the box's
load-only floor (boxes of the pool differ by 10 %), next to the defaults. One engine per setting, one process.
  python tools/period_sweep.py [REL ...]      (default: 0.99 .. 1.04)
Per setting: sustained median (us per query), the driver's command line (fresh engine, 5 queries of warm-up, ONE launch of 20) and a
32-query launch on its own."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402


def main():
    rels = [float(v) for v in sys.argv[1:]] or [0.99, 1.0, 1.005, 1.01, 1.015, 1.02, 1.03, 1.04]
    rows, cols, nnz = int(os.environ.get("ROWS", 1000000)), int(os.environ.get("COLS", 1024)), int(os.environ.get("NNZ", 20))
    torch.cuda.init()
    mod = _pkg.load()
    m = mod.generate_matrix(rows, cols, nnz, "gamma", 2)
    nq = 64
    xs = np.stack([mod.create_sample_vector(cols, True, False, True, 1000 + i) for i in range(nq)])
    dxs = torch.from_numpy(xs).cuda()

    def run(label):
        eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
        eng.time_queries(dxs.data_ptr(), nq, 5)
        torch.cuda.synchronize()
        drv = eng.time_queries(dxs.data_ptr(), nq, 20) / 1e3
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            eng.time_query_batches(dxs.data_ptr(), nq, 256, 8)
        reps = [v / 1e3 for v in eng.time_query_batches(dxs.data_ptr(), nq, 256, 26)][2:]
        one = sorted(eng.time_queries(dxs.data_ptr(), nq, 32) / 1e3 for _ in range(5))[2]
        floor = sorted(eng.time_stream_read(64) / 1e3 for _ in range(3))[1]
        c = eng.debug_counters()
        val, idx = eng.read_result()
        eng.close()
        print(json.dumps({"setting": label, "sustained_us": round(float(np.median(reps)), 2), "p95_over_median": round(float(np.percentile(reps, 95) / np.median(reps)), 3),
                          "driver_line_us": round(drv, 2), "one_launch_of_32_us": round(one, 2), "floor_us": round(floor, 2),
                          "checks_failed": c["checks_failed"], "pacing": f"{c['pace_quantum']}x{c['pace_levels']}/T{c['pace_period_ns']}"}), flush=True)
        return floor

    mod.set_option("PACE_PERIOD", None)
    floor = run("defaults")
    for r in rels:
        mod.set_option("PACE_PERIOD", str(int(round(floor * 1000 * r))))
        run(f"PACE_PERIOD = {r:.3f} x floor = {int(round(floor * 1000 * r))} ns")
    mod.set_option("PACE_PERIOD", None)
    run("defaults")


if __name__ == "__main__":
    main()
