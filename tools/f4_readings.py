"""SURVEY.md section 8 row f4, the reference's HLS approximations on BASELINE configs[2]'s matrix (10^6 x 1024, 20 per row, gamma):
precision@100 against the fp32 gold at 20 and 25 bits for the three readings experiments.hls_dataflow_topk offers --
  structural          overfull="count": exact W-bit scores, per-slot lists, unflushed last rows; overfull packets only counted
  cores, core ids     overfull="model": LIMITED_FINISHED_ROWS as spmv_bscsr_top_k_multicore.hpp:104-149,246-326 implements it
                      (dropped products, carried last segment, slipping row counter) -- equal to oracle/hls_model.c list for list
  cores, matrix ids   the same dataflow reporting the matrix row behind each offer (a row counter that does not slip)
beside the paper's 96.7-98.4 % at 20 bits. CPU only (the structural reading's scores come from the integer model here; on a GPU
they come from the engine's SpMV-only kernel, tests/test_gpu_partitions.py). Usage: python tools/f4_readings.py [queries]"""
import os
import sys
import time
from importlib import import_module

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg  # noqa: E402
import oracle_lib as oracle  # noqa: E402

pkg = _pkg.load()
ex = import_module("approximate_spmv_topk_amd.experiments")
n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m = pkg.generate_matrix(1000000, 1024, 20, "gamma", 2)
print(f"matrix 10^6 x 1024, {m.row.shape[0]} entries (gamma), 32 partitions x 4 lists x K = 8, k = 100, {n_q} queries")
for W in (20, 25):
    B = ex.bscsr_packet_size(W)
    res = {"structural": [], "cores, core ids": [], "cores, core ids (all candidates)": [], "cores, matrix ids": []}
    equal = True
    t0 = time.time()
    for q in range(n_q):
        x = pkg.create_sample_vector(1024, True, False, True, 11 + q)
        gold = set(oracle.gold_topk(m.row, m.col, m.val, x, 100)[0].tolist())
        y, _ = oracle.fixed_scores(m.row, m.col, m.val, x, m.rows, W)
        si, _, info = ex.hls_dataflow_topk(m.row, y, m.rows, 100, 32, 8, B, 4)
        ci, cv, minfo = ex.hls_dataflow_topk(m.row, None, m.rows, 4096, 32, 8, B, 4, overfull="model", col=m.col, val=m.val, vec=x, fixed_width=W)
        ti, _, _ = ex.hls_dataflow_topk(m.row, None, m.rows, 100, 32, 8, B, 4, overfull="model", col=m.col, val=m.val, vec=x, fixed_width=W, ids="matrix")
        oi, ov, _, _ = oracle.hls_model_topk(m.row, m.col, m.val, x, m.rows, 32, B, 8, 4, W)
        equal = equal and np.array_equal(oi, ci) and np.array_equal(ov.view(np.uint32), cv.view(np.uint32))
        res["structural"].append(len(set(si.tolist()) & gold) / 100)
        res["cores, core ids"].append(len(set(ci[:100].tolist()) & gold) / 100)
        res["cores, core ids (all candidates)"].append(len(set(ci.tolist()) & gold) / 100)
        res["cores, matrix ids"].append(len(set(ti.tolist()) & gold) / 100)
    print(f"{W} bits, {B} entries per packet, {info['overfull_packets']} overfull packets, {minfo['lost_rows']} rows never offered; "
          f"model == oracle/hls_model.c on every query: {equal}  ({time.time() - t0:.0f} s)")
    for name, v in res.items():
        print(f"   {name:34s} precision@100 mean {np.mean(v):.3f}  min {min(v):.2f}  max {max(v):.2f}   {v}")
