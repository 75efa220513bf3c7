#!/usr/bin/env python3
"""Timeline of the stream kernel from per-wave wall-clock stamps (TKSPMV_TRACE=1, tkspmv_debug_trace).

Runs a back-to-back sequence on the BASELINE workload, then prints, for the last launches, when the waves entered,
had x staged, finished their first packet, finished the loop, and left -- as percentiles over all streaming waves,
in microseconds relative to the first wave's entry of that launch -- plus the gap to the next launch.
Development tool; not part of the product path."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

os.environ["TKSPMV_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--cols", type=int, default=1024)
ap.add_argument("--nnz", type=int, default=20)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--replicas", type=int, default=4)
ap.add_argument("--iters", type=int, default=203)
a = ap.parse_args()

import torch  # noqa: E402

mod = _pkg.load()
from importlib import import_module  # noqa: E402
_lib = import_module("approximate_spmv_topk_amd._lib")
m = mod.generate_matrix(a.rows, a.cols, a.nnz, "gamma", 2)
xs = np.stack([mod.create_sample_vector(a.cols, True, False, True, 1000 + i) for i in range(8)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=0, stream_replicas=a.replicas)
info = eng.info()
grid = info["grid"]
eng.enqueue_many(dxs.data_ptr(), 8, a.iters)
eng.synchronize()
words = 4 * (grid + 1) * 9 * 8
buf = np.zeros(words, dtype=np.uint64)
got = C.c_uint64()
_lib.check(_lib.lib().tkspmv_debug_trace(eng._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), words, C.byref(got)))
t = buf.reshape(4, grid + 1, 9, 8).astype(np.int64)
# order the four slots by time
starts = [int(t[s, 1:grid, :8, 0][t[s, 1:grid, :8, 0] > 0].min()) for s in range(4)]
order = np.argsort(starts)
names = ["entry", "x staged", "first packet", "loop done", "deferred judged", "flush done"]
prev_end = None
for s in order:
    w = t[s, 1:grid, :8, :]  # streaming waves of blocks 1..grid-1 (block 0 is the deferred selection)
    live = w[..., 3] > 0
    base = w[..., 0][live].min()
    print(f"launch slot {s}: first entry at {base * 0.01:.2f} us"
          + (f", gap since previous launch's last exit {0.01 * (base - prev_end):.2f} us" if prev_end else ""))
    for j, nm in enumerate(names):
        v = (w[..., j][live] - base) * 0.01
        print(f"  {nm:16s} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  p50 {np.percentile(v, 50):6.2f}"
              f"  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
    sel = t[s, 0, :, :]
    if sel[0, 5] > 0:
        print(f"  selection workgroup: entry {0.01 * (sel[0, 0] - base):6.2f}  exit {0.01 * (sel[:, 5].max() - base):6.2f}")
    srv = t[s, 1:grid, 8, :]
    ok = srv[:, 5] > 0
    if ok.any():
        print(f"  server waves exit: p50 {np.percentile((srv[ok, 5] - base) * 0.01, 50):6.2f}  max {((srv[ok, 5] - base) * 0.01).max():6.2f}")
    prev_end = max(int(w[..., 5][live].max()), int(srv[ok, 5].max()) if ok.any() else 0)
    # structure of the end skew: inside workgroups or between them / between CUs?
    ld = np.where(live, w[..., 3], 0)
    wg_max = (ld.max(axis=1) - base) * 0.01
    wg_min = (np.where(live, w[..., 3], 1 << 62).min(axis=1) - base) * 0.01
    okwg = live.all(axis=1)
    print(f"  loop-done spread inside a workgroup: p50 {np.percentile((wg_max - wg_min)[okwg], 50):5.2f}  p90 {np.percentile((wg_max - wg_min)[okwg], 90):5.2f};"
          f"  slowest wave per workgroup: p10 {np.percentile(wg_max[okwg], 10):5.2f} p50 {np.percentile(wg_max[okwg], 50):5.2f} p90 {np.percentile(wg_max[okwg], 90):5.2f}")
    hw = w[:, 0, 6]
    xcc = (hw >> 32) & 0xF
    hwid = hw & 0xFFFFFFFF
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    cu_key = xcc * 1024 + se * 64 + sh * 16 + cu
    keys, counts = np.unique(cu_key, return_counts=True)
    print(f"  distinct CUs seen: {len(keys)}; workgroups per CU: " + ", ".join(f"{c}x{(counts == c).sum()}" for c in np.unique(counts)))
    cu_done = np.array([wg_max[cu_key == kk].max() for kk in keys])
    print(f"  slowest wave per CU: p10 {np.percentile(cu_done, 10):5.2f} p50 {np.percentile(cu_done, 50):5.2f} p90 {np.percentile(cu_done, 90):5.2f} max {cu_done.max():5.2f}")
    for c in np.unique(counts):
        sel_c = np.isin(cu_key, keys[counts == c])
        print(f"    CUs holding {c} workgroup(s): median workgroup finish {np.percentile(wg_max[sel_c], 50):5.2f}")
    print(f"  reducer workgroups (first 8 streaming blocks) finish: {wg_max[:8].round(2).tolist()}; slowest 8 blocks: {(np.argsort(wg_max)[-8:] ).tolist()} at {np.sort(wg_max)[-8:].round(2).tolist()}")
    ent = (w[:, 0, 0] - base) * 0.01
    late = ent > 1.5
    if late.any():
        print(f"  {late.sum()} workgroup(s) entered late (> 1.5 us): entry {ent[late].round(2).tolist()}, finish {wg_max[late].round(2).tolist()}")
    # per-XCD view of the exits (blockIdx % 8)
    ex = (w[..., 5] - base) * 0.01
    per_xcd = [np.percentile(ex[(np.arange(1, grid) % 8) == x][live[(np.arange(1, grid) % 8) == x]], 50) for x in range(8)]
    print("  median exit per XCD:", " ".join(f"{v:6.2f}" for v in per_xcd))
