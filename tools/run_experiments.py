#!/usr/bin/env python3
"""Experiment driver: the role of the reference's test_spmv_topk.py:66-111 for this engine.

For every (rows, cols, distribution, nnz/row) of the grid: make sure the MatrixMarket file exists in the matrix folder
(reference naming: matrix_{rows}_{cols}_{nnz}_{dist}.mtx; generated with this package's generator when missing), run
the drop-in executable with the reference's flags (-t NITER -k K -r), keep its CSV in the output folder, then print
the accuracy table (precision@t, Kendall's tau, NDCG against the executable's own CPU gold) and kernel times.
Needs a GPU. Example:
  python tools/run_experiments.py --rows 10000 100000 --cols 512 1024 --dist uniform gamma --nnz 20 40 -k 100 -t 30
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs="+", default=[10000, 100000])
ap.add_argument("--cols", type=int, nargs="+", default=[512, 1024])
ap.add_argument("--dist", nargs="+", default=["uniform", "gamma"])
ap.add_argument("--nnz", type=int, nargs="+", default=[20, 40])
ap.add_argument("-k", type=int, default=100)
ap.add_argument("-t", "--niter", type=int, default=30)
ap.add_argument("--matrix-folder", default=os.path.join(ROOT, "gpurun_out", "matrices"))
ap.add_argument("--out-folder", default=os.path.join(ROOT, "gpurun_out", "results", time.strftime("%Y_%m_%d_%H_%M_%S")))
ap.add_argument("--cache", action="store_true", help="keep packed matrices next to the MatrixMarket files (TKSPMV_CACHE_DIR)")
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--generate", action="store_true", help="no MatrixMarket files: the executable generates every matrix in memory "
                                                         "(TKSPMV_GENERATE; the reference's grid reaches 7 GB of text per matrix)")
ap.add_argument("--json", default=None, help="also write the table here (profiles/r05_paper_grid.json)")
ap.add_argument("--impl", type=int, default=0, help="engine variant passed as -i (0 streaming kernel, 1 row per lane, 2 scores + select)")
ap.add_argument("--bits", nargs="+", default=["f32"],
                help="value types to run: f32, f16 (the executable's -a) or a fixed-point width such as 20b, 25b, 32b "
                     "(the reference's FPGA builds, test_spmv_topk.py:42-47; TKSPMV_FIXED_WIDTH)")
a = ap.parse_args()

mod = _pkg.load()
from importlib import import_module  # noqa: E402
ex = import_module("approximate_spmv_topk_amd.experiments")

os.makedirs(a.matrix_folder, exist_ok=True)
os.makedirs(a.out_folder, exist_ok=True)
# matrices are written zero-based (the reference driver's ZERO_INDEXED = True, test_spmv_topk.py:26-35); "auto" also accepts
# one-based files left in the matrix folder by create_matrices.py
env = dict(os.environ, TKSPMV_SEED=str(a.seed), TKSPMV_INDEX_BASE="auto")
if a.cache:
    env["TKSPMV_CACHE_DIR"] = a.matrix_folder
grid = [(s, c, d, n, b) for s in a.rows for c in a.cols for d in a.dist for n in a.nnz for b in a.bits]
table = []
for i, (s, c, d, n, bits) in enumerate(grid):
    mtx = os.path.join(a.matrix_folder, ex.matrix_name(s, c, n, d))
    if not a.generate and not os.path.exists(mtx):
        mod.write_mtx(mtx, mod.generate_matrix(s, c, n, d, a.seed + i // len(a.bits)), index_base=0)
    out = os.path.join(a.out_folder, ex.result_name(s, c, d, n, a.k, a.niter, bits=bits))
    cmd = [ex.default_exe(), "-t", str(a.niter), "-m", mtx, "-k", str(a.k), "-i", str(a.impl), "-r"] + (["-a"] if bits == "f16" else [])
    run_env = dict(env, TKSPMV_FIXED_WIDTH=bits[:-1]) if bits.endswith("b") else dict(env)
    if a.generate:
        run_env["TKSPMV_GENERATE"] = f"{s},{c},{n},{d},{a.seed + i // len(a.bits)}"
    print(f"running {i + 1}/{len(grid)}: {' '.join(cmd)} > {out}", flush=True)
    t_run = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=run_env)
    if r.returncode != 0:
        print("  failed:", r.stderr.strip() or r.stdout.strip()[-300:])
        continue
    open(out, "w").write(r.stdout)
    acc = ex.accuracy(ex.read_result_csv(out), thresholds=[t for t in ex.THRESHOLDS if t <= a.k])
    acc.update(rows=s, cols=c, dist=d, nnz=n, k=a.k, bits=bits, file=os.path.basename(out), wall_s=round(time.time() - t_run, 1))
    table.append(acc)
    ts = [t for t in (1, 8, 50, 100) if f"prec_{t}" in acc]
    print("  " + "  ".join(f"prec@{t} {acc[f'prec_{t}']:.3f} tau@{t} {acc[f'kendall_{t}']:.3f} ndcg@{t} {acc[f'ndcg_{t}']:.4f}" for t in ts[-2:])
          + f"  hw_exec {acc['hw_exec_time_ms_mean'] * 1e3:.1f} us (+- {acc['hw_exec_time_ms_std'] * 1e3:.1f})  cpu gold top-k {acc['sw_topk_time_ms_mean']:.2f} ms")
json.dump(table, open(os.path.join(a.out_folder, "accuracy.json"), "w"), indent=1)
if a.json:
    prev = json.load(open(a.json)) if os.path.exists(a.json) else []
    json.dump(prev + table, open(a.json, "w"), indent=1)
print("results in", a.out_folder)
