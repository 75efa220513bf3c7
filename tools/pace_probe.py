"""Several ENGINE INSTANCES per setting, interleaved in one process (round 5: the same setting differs by up to 6 % from one engine
instance to the next on one box -- where the stream copies land in memory --, more than most changes are worth).
  [TKSPMV_LIB=...] python tools/pace_probe.py NAME INSTANCES "opt=val,opt=val" "opt=val" ...      (an empty string = the defaults)
Per setting: median / min / max over the instances of the sustained median (us per query), of the driver's command line (a fresh
engine, 5 queries of warm-up, ONE launch of 20) and of the load-only floor. ROWS COLS NNZ from the environment (default 1M x 1024 x 20).
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


def main():
    name, n_inst = sys.argv[1], int(sys.argv[2])
    settings = sys.argv[3:] or [""]
    rows, cols, nnz = int(os.environ.get("ROWS", 1000000)), int(os.environ.get("COLS", 1024)), int(os.environ.get("NNZ", 20))
    torch.cuda.init()
    mod = _pkg.load()
    m = mod.generate_matrix(rows, cols, nnz, "gamma", 2)
    nq = 64
    xs = np.stack([mod.create_sample_vector(cols, True, False, True, 1000 + i) for i in range(nq)])
    dxs = torch.from_numpy(xs).cuda()
    res = {s: {"sustained": [], "driver": [], "floor": []} for s in settings}
    touched = set()
    for inst in range(n_inst):
        for s in settings:
            for o in touched:
                mod.set_option(o, None)
            for kv in filter(None, s.split(",")):
                k, v = kv.split("=")
                mod.set_option(k, v)
                touched.add(k)
            eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
            eng.time_queries(dxs.data_ptr(), nq, 5)
            torch.cuda.synchronize()
            res[s]["driver"].append(eng.time_queries(dxs.data_ptr(), nq, 20) / 1e3)
            t_end = time.perf_counter() + 0.4
            while time.perf_counter() < t_end:
                eng.time_query_batches(dxs.data_ptr(), nq, 256, 8)
            reps = [v / 1e3 for v in eng.time_query_batches(dxs.data_ptr(), nq, 256, 26)][2:]
            res[s]["sustained"].append(float(np.median(reps)))
            res[s].setdefault("launch20", []).append(sorted(eng.time_queries(dxs.data_ptr(), nq, 20) / 1e3 for _ in range(9))[4])  # (ONE launch of 20, each behind a wait: median of 9)
            res[s]["floor"].append(sorted(eng.time_stream_read(64) / 1e3 for _ in range(3))[1])
            c = eng.debug_counters()
            res[s].setdefault("pace", []).append(f"{c.get('pace_quantum')}x{c.get('pace_levels')}/T{c.get('pace_period_ns')}" + (f"(tuned {c.get('pace_tuned_us')}us)" if c.get("pace_tuned_us") else ""))
            eng.close()
    for s in settings:
        r = res[s]
        print(json.dumps({"name": name, "setting": s or "defaults", "instances": n_inst,
                          "sustained_us": [round(float(np.median(r["sustained"])), 2), round(min(r["sustained"]), 2), round(max(r["sustained"]), 2)],
                          "driver_line_us": [round(float(np.median(r["driver"])), 2), round(min(r["driver"]), 2), round(max(r["driver"]), 2)],
                          "launch_of_20_us": [round(float(np.median(r["launch20"])), 2), round(min(r["launch20"]), 2), round(max(r["launch20"]), 2)],
                          "floor_us": round(float(np.median(r["floor"])), 2), "pace": r.get("pace")}))


if __name__ == "__main__":
    main()
