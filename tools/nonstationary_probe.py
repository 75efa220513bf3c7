"""The nonstationary leg of bench.py on its own (round 5): BASELINE configs[1], queries scaled by 1 / 0.01 / 3 at random, every 16th
sign-flipped and every 16th concentrated on 32 columns; stationary stream beside it, same engine. Parity of the last query of
every leg against the packed-order oracle.
  [TKSPMV_SIGNATURES=0] python tools/nonstationary_probe.py [INSTANCES]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg  # noqa: E402


def main():
    n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.cuda.init()
    mod = _pkg.load()
    import oracle_lib as oracle
    m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
    nq, k = 64, 100
    xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(nq)])
    rng = np.random.default_rng(11)
    sc = rng.choice([1.0, 0.01, 3.0], size=nq).astype(np.float32)
    xs_ns = xs * sc[:, None]
    xs_ns[5::16] *= np.float32(-1.0)
    for j in range(11, nq, 16):
        keep = rng.choice(1024, size=32, replace=False)
        mask = np.zeros(1024, dtype=bool)
        mask[keep] = True
        xs_ns[j, ~mask] = 0.0
    xs_ns = np.ascontiguousarray(xs_ns.astype(np.float32))
    dxs, dns = torch.from_numpy(xs).cuda(), torch.from_numpy(xs_ns).cuda()
    for inst in range(n_inst):
        eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=k, device=0, stream_replicas=4)
        info = eng.info()
        C = info["packet_entries"] // 64
        packed = mod.Packed(m, k=k, nnz_per_lane=C, n_wave_partitions=(info["batch_mode"] >> 16) or info["n_wave_partitions"])
        raw = packed.raw()
        out = {"instance": inst, "signatures": os.environ.get("TKSPMV_SIGNATURES", "1")}
        for name, d, host in (("stationary", dxs, xs), ("nonstationary", dns, xs_ns), ("stationary_again", dxs, xs)):
            c0 = eng.debug_counters()
            eng.time_queries(d.data_ptr(), nq, 64)
            reps = [v / 1e3 for v in eng.time_query_batches(d.data_ptr(), nq, 256, 10)][2:]
            val, idx = eng.read_result()
            c1 = eng.debug_counters()
            x_last = host[(256 - 1) % nq]
            yp, present = oracle.packed_scores(raw, x_last, m.rows, C)
            ei, ev = oracle.select_topk(yp, present, k, 0.0)
            out[name] = {"us_median": round(float(np.median(reps)), 2), "us_max": round(max(reps), 2),
                         "checks_failed": c1["checks_failed"] - c0["checks_failed"], "late_repairs": c1["late_repairs"] - c0["late_repairs"],
                         "suspended_for_after": c1["suspended_for"], "exact": bool(np.array_equal(idx, ei) and np.array_equal(val.view(np.uint32), ev.view(np.uint32)))}
        print(json.dumps(out))
        eng.close()


if __name__ == "__main__":
    main()
