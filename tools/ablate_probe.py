"""Batch kernel under the ablation switches of TKSPMV_DBG_FLAGS (DBG instantiation: 2: candidate path off -- wrong
results, timing only; 4: no threshold duty; 16: every state set keeps its final threshold and the same vector comes back
to it: an exact threshold from a query's first packet), one engine per setting; `None` = the production instantiation,
with single launches and the SpMV-only kernel beside it (MULTI=1,4,8 adds the multi-query path; STATS=1 the counters).
  python tools/ablate_probe.py ROWS COLS NNZ [flags ... | none]
Round 4 removed the ablation switches from the library (what they showed is in DESIGN.md section 9): in this tree only `none` --
the production kernel beside its load-only floor, which tools/ab_variants.sh runs interleaved with an older tree under _ab/ --
does anything; the flags still work when the probe runs inside such an older tree."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

torch.cuda.init()
mod = _pkg.load()
rows, cols, nnz = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1000000, 1024, 20)
m = mod.generate_matrix(rows, cols, nnz, "gamma", 2)
xs = np.stack([mod.create_sample_vector(cols, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
if os.environ.get("C16"):
    os.environ["TKSPMV_F32_C12"] = "0"  # 16-bit column words (A/B against the 12-bit default)
settings = [None] if sys.argv[4:5] == ["none"] else [None] + [int(f) for f in (sys.argv[4:] or ["0", "2", "16"])] + [None]
for flags in settings:
    if flags is None:
        os.environ.pop("TKSPMV_DBG_FLAGS", None)
        os.environ.pop("TKSPMV_STATS", None)
    else:
        os.environ["TKSPMV_DBG_FLAGS"] = str(flags)
        if os.environ.get("STATS"):
            os.environ["TKSPMV_STATS"] = "1"
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4,
                   nnz_per_lane=int(os.environ.get("NNZ_PER_LANE", "0")), threads_per_wg=int(os.environ.get("THREADS_PER_WG", "0")),
                   waves_per_cu=int(os.environ.get("WAVES_PER_CU", "0")))
    eng.enqueue_many(dxs.data_ptr(), 64, 256)
    eng.synchronize()
    nx = 32 if (flags is not None and flags & 16) else 64  # (flag 16: the same vector must come back to the same state set)
    eng.enqueue_many(dxs.data_ptr(), nx, 64)
    t = sorted(eng.time_queries(dxs.data_ptr(), nx, 512) / 1e3 for _ in range(7))[3]
    r = sorted(eng.time_stream_read(64) / 1e3 for _ in range(5))[2] if hasattr(eng, "time_stream_read") else float("nan")
    extra = ""
    if flags is not None and os.environ.get("STATS"):
        p = eng.profile(dxs.data_ptr(), 64, 128)
        extra = f"; slow-path packets {p['slow_paths_avg']:.0f}, rows offered {p['appended_avg']:.0f}, candidates {p['candidates_avg']:.0f} per query"
    if flags is None:
        p = eng.profile(dxs.data_ptr(), 64, 200)
        extra = (f"; single launches: {p['stream_kernel_ns'] / 1e3:.2f} us (event bracket, empty kernel {p['event_bracket_ns'] / 1e3:.2f}), "
                 f"SpMV-only {p['scores_kernel_ns'] / 1e3:.2f}")
    if flags is None and os.environ.get("MULTI"):
        for mq in (int(v) for v in os.environ["MULTI"].split(",")):
            me = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4, multi_q=mq)
            me.time_multi(dxs.data_ptr(), 64, 256)
            tm = sorted(me.time_multi(dxs.data_ptr(), 64, 1024) / 1e3 for _ in range(5))[2]
            extra += f"; multi_q={mq}: {tm:.2f} us per query"
            me.close()
    print(f"{rows}x{cols}x{nnz} flags {flags}: {t:.2f} us per query (read-only floor {r:.2f}){extra}")
    eng.close()
