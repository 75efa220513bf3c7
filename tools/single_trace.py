"""Per-wave timeline of single_kernel launches (TKSPMV_TRACE=1): entry, x staged, first packet reduced, loop done, record delivered;
the selecting workgroup's end. Development probe."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["TKSPMV_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402
from importlib import import_module  # noqa: E402

mod = _pkg.load()
_lib = import_module("approximate_spmv_topk_amd._lib")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(8)])
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=4)
grid = eng.info()["grid"]
for i in range(12):
    eng.reset(xs[i % 8])
    ns = eng()
print("tkspmv_run device time of the last launch: %.2f us" % (ns / 1e3), eng.debug_counters())
tw = (grid + 1) * 9 * 8
words = 4 * tw
buf = np.zeros(words, dtype=np.uint64)
got = C.c_uint64()
_lib.check(_lib.lib().tkspmv_debug_trace(eng._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), words, C.byref(got)))
names = ["entry", "x staged", "first packet reduced", "loop done", "record delivered / ticket"]
for s in range(4):
    t = buf[s * tw: s * tw + grid * 8 * 8].reshape(grid, 8, 8).astype(np.int64)
    live = t[..., 3] > 0
    if not live.any():
        continue
    base = t[..., 0][t[..., 0] > 0].min()
    print(f"slot {s}: {int(live.sum())} streaming waves (us since the first wave's entry)")
    for j, nm in enumerate(names):
        v = t[..., j][live] if j < 4 else t[:, 0, 4][t[:, 0, 4] > 0]
        v = (v - base) * 0.01
        print(f"  {nm:28s} min {v.min():6.2f} p10 {np.percentile(v,10):6.2f} p50 {np.percentile(v,50):6.2f} p90 {np.percentile(v,90):6.2f} p99 {np.percentile(v,99):6.2f} max {v.max():6.2f}")
    for j, nm in ((5, "workgroup staged (wave 0)"), (6, "8 best chosen"), (7, "record drained")):
        v = (t[:, 0, j][t[:, 0, j] > 0] - base) * 0.01
        print(f"  {nm:28s} min {v.min():6.2f} p10 {np.percentile(v,10):6.2f} p50 {np.percentile(v,50):6.2f} p90 {np.percentile(v,90):6.2f} p99 {np.percentile(v,99):6.2f} max {v.max():6.2f}")
    ss = (buf[s * tw + grid * 64: s * tw + grid * 64 + 8].astype(np.int64) - base) * 0.01
    print("  selection: start %.2f | loads returned %.2f | first cut %.2f | keys in LDS %.2f | ranked %.2f | host stores drained %.2f | end %.2f" % (ss[7], ss[0], ss[1], ss[2], ss[3], ss[4], ss[5]))
    # by XCD (blockIdx % 8) and by CU round (first 256 workgroups vs the rest)
    ld = np.where(live, (t[..., 3] - base) * 0.01, np.nan)
    print("  loop done p50 by blockIdx % 8:", [round(float(np.nanmedian(ld[x::8])), 2) for x in range(8)])
    print("  loop done p50 first 256 workgroups / rest:", round(float(np.nanmedian(ld[:256])), 2), round(float(np.nanmedian(ld[256:])), 2))
    print("  loop done p50 by wave index:", [round(float(np.nanmedian(ld[:, w])), 2) for w in range(8)])
    dur = np.where(live, (t[..., 3] - t[..., 2]) * 0.01, np.nan)
    print("  first packet -> loop done: p10 %.2f p50 %.2f p90 %.2f" % tuple(np.nanpercentile(dur, [10, 50, 90])))
eng.close()
