"""Which launches of a back-to-back series are the slow ones, and what do their workgroups do differently? (option WG_TIMES keeps the
stamps of the LAST launch: series of 9..16 launches, the last one's duration and stamps kept.)   python tools/spike_probe.py [SERIES]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TKSPMV_WG_TIMES"] = "1"
import _pkg  # noqa: E402

n_series = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.cuda.init()
mod = _pkg.load()
from importlib import import_module
_lib = import_module(mod.__name__ + "._lib")
m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
grid = eng.info()["grid"]
eng.time_query_batches(dxs.data_ptr(), 64, 32, 64)
out = []
for i in range(n_series):
    v = [x / 1e3 for x in eng.time_query_batches(dxs.data_ptr(), 64, 32, 9 + i % 8)]
    words = 33 * grid
    buf = np.zeros(words, dtype=np.uint64)
    got = C.c_uint64()
    _lib.check(_lib.lib().tkspmv_debug_trace(eng._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), words, C.byref(got)))
    raw = buf.reshape(33, grid)
    t = (raw & np.uint64(0x00FFFFFFFFFFFFFF)).astype(np.int64)
    U = (raw >> np.uint64(56)).astype(np.int64)
    n_wg = grid - 4
    E = t[32, :n_wg]
    t0 = E.min()
    T = (t[:32, :n_wg] - t0) * 0.01
    D = np.diff(np.vstack([(E - t0)[None, :] * 0.01, T]), axis=0)
    sel_done = (t[:32, n_wg + 1] - t0) * 0.01
    out.append({"us_per_query": v[-1], "series": [round(x, 1) for x in v], "entry_last": float((E - t0).max() * 0.01), "x0": float(np.median(U[32, :n_wg]) * 0.1),
                "q0_median": float(np.median(T[0])), "per_query_median": [round(float(np.median(D[q])), 1) for q in range(32)],
                "per_query_p95": [round(float(np.percentile(D[q], 95)), 1) for q in range(32)], "last_handover": float(T[31].max()), "median_handover_last": float(np.median(T[31])),
                "last_selection_done": float(sel_done[31]), "slowest_wg_total": float((T[31] - (E - t0) * 0.01).max()),
                "laggards": [(int(w), [round(float(x), 0) for x in D[:, w]], [int(u) for u in U[:32, w]]) for w in np.argsort(T[31])[-4:]],
                "n_late_10us": int((T[31] > np.median(T[31]) + 10).sum())})
out.sort(key=lambda o: o["us_per_query"])
for o in out[:3] + out[-4:]:
    print(f"{o['us_per_query']:.2f} us/q | series {o['series']} | last entry {o['entry_last']:.1f} x0 {o['x0']:.1f} q0 {o['q0_median']:.1f} | median hand-over of the last query {o['median_handover_last']:.1f}, "
          f"last {o['last_handover']:.1f}, last selection {o['last_selection_done']:.1f}")
    print("     per-query median:", o["per_query_median"])
    print("     per-query p95   :", o["per_query_p95"])
    print(f"     workgroups more than 10 us behind the median at the end: {o['n_late_10us']}")
    for w, d, u in o["laggards"]:
        print(f"     wg {w} (xcd {w % 8}, cu slot {w // 8}): durations {d}")
        print(f"          pauses by rank chosen {u}")
eng.close()
