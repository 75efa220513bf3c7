"""Who leads, who lags: the hand-over time of every workgroup for every query of ONE batch launch (option WG_TIMES; the production
kernel of local thresholds stamps its tickets), after a sustained warm-up.
  [TKSPMV_PACE=.. TKSPMV_PACE_LEVELS=..] python tools/wg_times.py OUT.npz [N_Q]
Prints per query: first / median / last hand-over and the spread; the per-query duration of the workgroups by XCD (bid % 8) and by
age on the CU (first-dispatched half against the second); how persistent a workgroup's place in the field is (rank correlation of
consecutive queries); and saves the raw [n_q][n_wg] matrix (us from the first workgroup's entry).
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TKSPMV_WG_TIMES"] = "1"
import _pkg  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    rows, cols, nnz = int(os.environ.get("ROWS", 1000000)), int(os.environ.get("COLS", 1024)), int(os.environ.get("NNZ", 20))
    torch.cuda.init()
    mod = _pkg.load()
    from importlib import import_module
    _lib = import_module(mod.__name__ + "._lib")
    m = mod.generate_matrix(rows, cols, nnz, "gamma", 2)
    xs = np.stack([mod.create_sample_vector(cols, True, False, True, 1000 + i) for i in range(64)])
    dxs = torch.from_numpy(xs).cuda()
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
    grid = eng.info()["grid"]
    for _ in range(20):
        eng.time_query_batches(dxs.data_ptr(), 64, 256, 8)
    us = eng.time_queries(dxs.data_ptr(), 64, n_q) / 1e3  # ONE launch of n_q queries: the stamps read below are its
    words = 33 * grid
    buf = np.zeros(words, dtype=np.uint64)
    got = C.c_uint64()
    _lib.check(_lib.lib().tkspmv_debug_trace(eng._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), words, C.byref(got)))
    raw = buf.reshape(33, grid)
    U = (raw >> np.uint64(56)).astype(np.int64)  # the pause (units per packet) each workgroup chose at that hand-over
    t = (raw & np.uint64(0x00FFFFFFFFFFFFFF)).astype(np.int64)
    n_sel = int((t[32] == 0).sum())  # (rows are indexed by streaming workgroup; the last n_selectors columns of a query's row hold the selector's stamps)
    n_wg = grid - n_sel
    entry = t[32, :n_wg]
    x0 = U[32, :n_wg] * 0.1  # the entry row's top byte: how long the first x took to stage (us)
    sel_seen, sel_done = t[:n_q, n_wg], t[:n_q, n_wg + 1]  # per query: all tickets seen by its selector / its selection done
    t0 = entry[entry > 0].min()
    T = (t[:n_q, :n_wg] - t0) * 0.01  # us
    E = (entry - t0) * 0.01
    print(f"{rows} x {cols}, {nnz} nnz/row; one launch of {n_q} queries: {us:.2f} us per query by the event pair; {n_wg} streaming workgroups; "
          f"PACE={os.environ.get('TKSPMV_PACE', 'default')} LEVELS={os.environ.get('TKSPMV_PACE_LEVELS', 'default')} BASE={os.environ.get('TKSPMV_PACE_BASE', '0')}")
    print(f"entry of the workgroups: median {np.median(E):.2f}, last {E.max():.2f} us")
    print(" q   first  median    last  spread | per-query duration: median   p5    p95 | by XCD (bid % 8) median duration                          | older half / younger half")
    prev = np.tile(E, (1, 1))[0]
    D = np.zeros_like(T)
    for q in range(n_q):
        D[q] = T[q] - (T[q - 1] if q else E)
    for q in range(n_q):
        d = D[q]
        xcd = " ".join(f"{np.median(d[np.arange(n_wg) % 8 == x]):5.2f}" for x in range(8))
        half = n_wg // 2
        print(f"{q:2d} {T[q].min():7.2f} {np.median(T[q]):7.2f} {T[q].max():7.2f} {T[q].max() - T[q].min():7.2f} | {np.median(d):26.2f} {np.percentile(d, 5):5.2f} {np.percentile(d, 95):5.2f} | {xcd} | "
              f"{np.median(d[:half]):5.2f} / {np.median(d[half:]):5.2f}")
    # persistence of a workgroup's place: Spearman correlation of the durations of consecutive queries
    def rank(v):
        r = np.empty_like(v)
        r[np.argsort(v)] = np.arange(len(v))
        return r
    cors = [float(np.corrcoef(rank(D[q]), rank(D[q + 1]))[0, 1]) for q in range(2, n_q - 1)]
    cors2 = [float(np.corrcoef(rank(D[q]), rank(D[q + 2]))[0, 1]) for q in range(2, n_q - 2)]
    print(f"rank correlation of a workgroup's duration in consecutive queries: median {np.median(cors):+.2f} (two apart: {np.median(cors2):+.2f})")
    tot = T[n_q - 1] - E
    print(f"whole launch per workgroup: median {np.median(tot):.1f}, min {tot.min():.1f}, max {tot.max():.1f} us; the launch waits {tot.max() - np.median(tot):.1f} us for its last workgroup")
    print(f"first x staged after the workgroup's entry: median {np.median(x0):.1f}, p95 {np.percentile(x0, 95):.1f}, max {x0.max():.1f} us")
    if n_sel >= 2 and sel_seen[n_q - 1] > 0:
        ls = (sel_seen - t0) * 0.01
        ld = (sel_done - t0) * 0.01
        print("selections: last ticket -> seen by the selector -> done (us after the last hand-over of the query): " +
              " ".join(f"{ls[q] - T[q].max():.1f}/{ld[q] - T[q].max():.1f}" for q in range(n_q)))
        print(f"the launch's last selection is done at {ld[n_q - 1]:.2f} us; event pair: {us * n_q:.2f} us")
    Un = U[:n_q, :n_wg]
    print("pause chosen at the hand-over (units per packet), by query: median / p90 / max:", " ".join(f"{int(np.median(Un[q]))}/{int(np.percentile(Un[q], 90))}/{int(Un[q].max())}" for q in range(n_q)))
    if out:
        np.savez_compressed(out, T=T, E=E, U=Un, us_per_query=us)
    eng.close()


if __name__ == "__main__":
    main()
