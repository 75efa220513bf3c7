"""Quick GPU probe (development aid): builds the cfg-2 engine, checks one query against the oracle, prints timings."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import _pkg
import oracle_lib as O

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--cols", type=int, default=1024)
ap.add_argument("--nnz", type=int, default=20)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--waves_per_cu", type=int, default=0)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--lane", type=int, default=0)
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--min_score", type=float, default=0.0)
ap.add_argument("--replicas", type=int, default=0)
a = ap.parse_args()
mod = _pkg.load()
t0 = time.time()
m = mod.generate_matrix(a.rows, a.cols, a.nnz, "gamma", 2)
t1 = time.time()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=0, waves_per_cu=a.waves_per_cu,
               threads_per_wg=a.threads, nnz_per_lane=a.lane, min_score=a.min_score, stream_replicas=a.replicas)
t2 = time.time()
info = eng.info()
print("gen %.2fs setup %.2fs" % (t1 - t0, t2 - t1), json.dumps(info))
nq = 64
xs = np.stack([mod.create_sample_vector(a.cols, True, False, True, 1000 + i) for i in range(nq)])
dxs = torch.from_numpy(xs).cuda()
if a.check:
    eng.reset(xs[0]); eng(); val, idx = eng.read_result()
    gi, gv = O.gold_topk(m.row, m.col, m.val, xs[0], a.k)
    print("set equal:", set(idx.tolist()) == set(gi.tolist()), "order equal:", np.array_equal(idx, gi),
          "max rel:", float(np.max(np.abs(val - gv) / gv)))
for rep in range(3):
    t = eng.profile(dxs.data_ptr(), nq, a.iters)
    alg = info["algorithmic_bytes"]
    print("scores-only %.2f us slow %.0f app %.0f |" % (t["scores_kernel_ns"] / 1e3, t["slow_paths_avg"], t["appended_avg"]), end=" ")
    print("query %.2f us  stream %.2f us  select %.2f us  cand %.1f | %.0f q/s  alg %.1f MB -> %.0f GB/s (%.1f%% of 8TB/s)"
          % (t["query_ns"] / 1e3, t["stream_kernel_ns"] / 1e3, t["select_kernel_ns"] / 1e3, t["candidates_avg"],
             1e9 / t["query_ns"], alg / 1e6, alg / t["query_ns"], alg / t["query_ns"] / 80))
# host-boundary rate: reset (H2D 4 KiB) + run (sync) + read_result (D2H 800 B) per query
import time as _t
eng.reset(xs[0]); eng(); eng.read_result()
t0 = _t.perf_counter()
for i in range(300):
    eng.reset(xs[i % nq]); eng(); eng.read_result()
print("host-boundary (set_query + run + read) per query: %.1f us" % ((_t.perf_counter() - t0) / 300 * 1e6))
