import os, sys, ctypes as C
os.environ["TKSPMV_TRACE"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
import torch
from importlib import import_module
mod = _pkg.load()
_lib = import_module("approximate_spmv_topk_amd._lib")
m = mod.generate_matrix(int(os.environ.get("ROWS", "1000000")), 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(8)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=4)
grid = eng.info()["grid"]
if os.environ.get("WARM"):  # sustained state first (carried thresholds, pacing, clocks): the traced launch is then a launch like any other
    for _ in range(int(os.environ["WARM"])):
        eng.enqueue_many(dxs.data_ptr(), 8, 256)
    eng.synchronize()
eng.enqueue_many(dxs.data_ptr(), 8, int(os.environ.get('NQ', '3')))
eng.synchronize()
words = 4 * (grid + 1) * 9 * 8
buf = np.zeros(words, dtype=np.uint64)
got = C.c_uint64()
_lib.check(_lib.lib().tkspmv_debug_trace(eng._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), words, C.byref(got)))
t = buf.reshape(4, grid + 1, 9, 8).astype(np.int64)
if os.environ.get("TRACE_NPZ"):
    np.savez_compressed(os.environ["TRACE_NPZ"], t=t)
slot = int(np.argmax(t[:, 1:, 0, 0].max(axis=1)))
w = t[slot]
base = w[1:, :, 0][w[1:, :, 0] > 0].min()
print("slot", slot, "selector stamps (us):", ((w[0, 0, :8] - base) * 0.01).round(1).tolist())
sel = w[0].reshape(-1)[8:16]
if sel[0] > 0 and sel[7] > 0:  # the phases of the selection of query 4 (s_memtime stamps rebased on the pair taken at its start)
    mt = lambda v: (sel[0] + (v - sel[1]) - base) * 0.01
    print(f"selection of query 4: poll from {(sel[0] - base) * 0.01:.1f}, ticket seen {(sel[2] - base) * 0.01:.1f}, loads returned {mt(sel[4]):.1f}, "
          f"keys in LDS {mt(sel[5]):.1f}, ranked {mt(sel[6]):.1f}, done {(sel[7] - base) * 0.01:.1f}")
names = ["entry", "qF start", "qM start", "qL start", "qF end", "qM end", "qL end"]
st = w[1:, :8, :]
for j, nm in enumerate(names):
    v = (st[..., j][st[..., j] > 0] - base) * 0.01
    if len(v): print(f"waves  {nm:9s} n={len(v):5d} min {v.min():7.1f} p1 {np.percentile(v,1):7.1f} p10 {np.percentile(v,10):7.1f} p50 {np.percentile(v,50):7.1f} p90 {np.percentile(v,90):7.1f} p99 {np.percentile(v,99):7.1f} max {v.max():7.1f}")
sv = w[1:, 8, :]
for j, nm in enumerate(["entry", "q0 staged", "q1 staged", "q2 staged", "q0 final", "q1 final", "q2 final"]):
    v = (sv[:, j][sv[:, j] > 0] - base) * 0.01
    if len(v): print(f"server {nm:9s} n={len(v):5d} min {v.min():9.1f} p50 {np.percentile(v,50):9.1f} p99 {np.percentile(v,99):9.1f} max {v.max():9.1f}")
late = np.argsort(sv[:, 5])[-5:]
print("slowest q1-final workgroups:", late.tolist(), ((sv[late, 5] - base) * 0.01).round(1).tolist())
for b in late[-2:]:
    print(" wg", b, "wave q1 end:", ((st[b, :, 5] - base) * 0.01).round(1).tolist(), "q1 start:", ((st[b, :, 2] - base) * 0.01).round(1).tolist())

# drift between the two workgroups of a CU-slot round (blocks 1..256 vs 257..511) per query
for j, nm in ((4, "q0 end"), (5, "q1 end"), (6, "q2 end")):
    a = st[:256, :, j]; b = st[256:, :, j]
    a = (a[a > 0] - base) * 0.01; b = (b[b > 0] - base) * 0.01
    print(f"{nm}: first-round workgroups p50 {np.percentile(a,50):7.1f}, second-round p50 {np.percentile(b,50):7.1f}")

# the middle query and its successor, wave by wave: how long each takes and what lies between them
ok = (st[..., 2] > 0) & (st[..., 5] > 0) & (st[..., 3] > 0) & (st[..., 6] > 0)
d1 = (st[..., 5] - st[..., 2])[ok] * 0.01
gap = (st[..., 3] - st[..., 5])[ok] * 0.01
d2 = (st[..., 6] - st[..., 3])[ok] * 0.01
per = (st[..., 3] - st[..., 2])[ok] * 0.01
for nm, v in (("qM duration (x ready -> counted out)", d1), ("gap qM counted out -> qM+1 x ready", gap), ("qM+1 duration", d2), ("qM start -> qM+1 start", per)):
    print(f"{nm:40s} mean {v.mean():6.2f} p10 {np.percentile(v,10):6.2f} p50 {np.percentile(v,50):6.2f} p90 {np.percentile(v,90):6.2f} p99 {np.percentile(v,99):6.2f} max {v.max():6.2f}")
dur = np.where(ok, (st[..., 5] - st[..., 2]) * 0.01, np.nan)
print("qM duration by wave index (mean over workgroups):", np.nanmean(dur, axis=0).round(2).tolist())
per_w = np.where(ok, (st[..., 3] - st[..., 2]) * 0.01, np.nan)
print("qM start -> qM+1 start by wave index:", np.nanmean(per_w, axis=0).round(2).tolist())
s0 = (st[..., 2][ok] - base) * 0.01
print(f"spread of qM starts over the waves: p1 {np.percentile(s0,1):.1f} p50 {np.percentile(s0,50):.1f} p99 {np.percentile(s0,99):.1f}")

sv7 = st[..., 7]
for slot_i, nm in enumerate(("qF", "qM", "qL")):
    surv = (sv7 >> (16 * slot_i + 8)) & 0xFF
    ftp = (sv7 >> (16 * slot_i)) & 0xFF
    notau = (sv7 >> (48 + slot_i)) & 1
    live = st[..., 4 + slot_i] > 0
    print(f"{nm}: survivors per wave at flush: mean {surv[live].mean():6.1f} p50 {np.percentile(surv[live],50):5.0f} p90 {np.percentile(surv[live],90):5.0f} max {surv[live].max()};"
          f" first packet with a threshold: p10 {np.percentile(ftp[live],10):4.0f} p50 {np.percentile(ftp[live],50):4.0f} p90 {np.percentile(ftp[live],90):4.0f} (255 = never);"
          f" waves with NO threshold at flush: {int(notau[live].sum())} of {int(live.sum())}; first-round mean {surv[:256][live[:256]].mean():6.1f} second-round mean {surv[256:][live[256:]].mean():6.1f}")

# server of the middle query: staged, first duty iteration, first threshold, finalised; iterations of duty
sv = w[1:, 8, :]
ok = sv[:, 5] > 0
stg_t = (sv[ok, 2] - base) * 0.01; fd = (sv[ok, 3] - base) * 0.01; ft = np.where(sv[ok, 7] > 0, (sv[ok, 7] - base) * 0.01, np.nan); fin = (sv[ok, 5] - base) * 0.01
print(f"middle query, servers: staged p50 {np.percentile(stg_t,50):7.1f}  first duty p50 {np.percentile(fd,50):7.1f}  first threshold p10 {np.nanpercentile(ft,10):7.1f} p50 {np.nanpercentile(ft,50):7.1f} p90 {np.nanpercentile(ft,90):7.1f} (none: {int(np.isnan(ft).sum())})  finalised p50 {np.percentile(fin,50):7.1f}; duty iterations p50 {np.percentile(sv[ok,0],50):4.0f}")
print("reducers:", "first threshold", ((sv[:8, 7] - base) * 0.01).round(1).tolist(), "iters", sv[:8, 0].tolist())
