"""One rung of the attribution ladder of the headline kernel (tools/ladder.sh runs it once per build of the library, interleaved on
ONE box): BASELINE configs[1], four rotating stream copies, the kernel under SUSTAINED load (every repetition enqueued before the
first wait), its load-only floor from the same process, and what the driver's command line times (20 queries behind 5).
  TKSPMV_LIB=_ab/L2/approximate-spmv-topk_amd/libtkspmv.so python tools/ladder_probe.py NAME [ROWS COLS NNZ]
Prints ONE JSON line. Rungs below 4 return no results (timing-only builds): nothing here looks at a result.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402


def pct(v, p):
    return float(np.percentile(np.asarray(v, dtype=np.float64), p))


def pci_address():
    """PCI address of the GPU this process runs on (matches the card list in tools/clock_sampler.py's header)."""
    try:
        p = torch.cuda.get_device_properties(0)
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:  # noqa: BLE001
        return None


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "?"
    rows, cols, nnz = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (1000000, 1024, 20)
    torch.cuda.init()
    mod = _pkg.load()
    m = mod.generate_matrix(rows, cols, nnz, "gamma", 2)
    nq = 64
    xs = np.stack([mod.create_sample_vector(cols, True, False, True, 1000 + i) for i in range(nq)])
    dxs = torch.from_numpy(xs).cuda()
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
    info = eng.info()
    # the driver's command line first (a fresh engine, 5 queries of warm-up, ONE launch of 20)
    eng.time_queries(dxs.data_ptr(), nq, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    drv = eng.time_queries(dxs.data_ptr(), nq, 20) / 1e3
    torch.cuda.synchronize()
    drv_host = (time.perf_counter() - t0) * 1e6 / 20
    # sustained: ~1.5 s of back-to-back launches before anything is kept
    t_end = time.perf_counter() + 1.5
    while time.perf_counter() < t_end:
        eng.time_query_batches(dxs.data_ptr(), nq, 256, 8)
    reps = [v / 1e3 for v in eng.time_query_batches(dxs.data_ptr(), nq, 256, 42)][2:]
    floor = sorted(eng.time_stream_read(64) / 1e3 for _ in range(7))[3]
    reps2 = [v / 1e3 for v in eng.time_query_batches(dxs.data_ptr(), nq, 256, 42)][2:]
    c = eng.debug_counters()
    out = {"name": name, "lib": os.environ.get("TKSPMV_LIB", "in-tree"), "rows": rows, "cols": cols, "nnz_per_row": nnz,
           "sustained_median_us": pct(reps + reps2, 50), "sustained_p95_us": pct(reps + reps2, 95),
           "p95_over_median": pct(reps + reps2, 95) / pct(reps + reps2, 50), "sustained_min_us": min(reps + reps2),
           "read_only_us": floor, "driver_line_kernel_us": drv, "driver_line_host_us": drv_host,
           "algorithmic_bytes": int(info["algorithmic_bytes"]),
           "frac_at_median": info["algorithmic_bytes"] / (pct(reps + reps2, 50) * 1e3) / 8000.0,
           "checks_failed": c.get("checks_failed"), "pace": os.environ.get("TKSPMV_PACE", "default"),
           "pace_in_force": f"{c.get('pace_quantum')}x{c.get('pace_levels')}", "pace_tuned_us": c.get("pace_tuned_us"), "pci": pci_address()}
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
