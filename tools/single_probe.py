#!/usr/bin/env python3
"""Where does the time of ONE query go (tkspmv_run: the reference loop's reset -> operator() -> read_result)?

Prints, for BASELINE configs[1] with rotating stream copies:
  * the hipEvent bracket around single fused launches (median / p95), and the same bracket around an empty kernel;
  * the per-wave timeline of single launches (TKSPMV_TRACE=1): entry, x staged, first packet, loop done, flush done;
  * the selection tail's shader-clock stamps (TKSPMV_STAMPS=1);
  * the host-boundary round trip (set_query + run + read).
Development tool; `--plain N` only runs N single queries (for rocprofv3 --kernel-trace)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--cols", type=int, default=1024)
ap.add_argument("--nnz", type=int, default=20)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--replicas", type=int, default=4)
ap.add_argument("--plain", type=int, default=0)
ap.add_argument("--impl", type=int, default=0)
a = ap.parse_args()

import torch  # noqa: E402

mod = _pkg.load()
from importlib import import_module  # noqa: E402
_lib = import_module("approximate_spmv_topk_amd._lib")
m = mod.generate_matrix(a.rows, a.cols, a.nnz, "gamma", 2)
xs = np.stack([mod.create_sample_vector(a.cols, True, False, True, 1000 + i) for i in range(8)])
dxs = torch.from_numpy(xs).cuda()


def engine(**env):
    for k_, v in env.items():
        os.environ[k_] = v
    e = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=0, stream_replicas=a.replicas, impl=a.impl)
    for k_ in env:
        del os.environ[k_]
    return e


def pct(v, name):
    v = np.asarray(v, dtype=np.float64)
    print(f"  {name:34s} min {v.min():7.2f}  p50 {np.percentile(v, 50):7.2f}  p95 {np.percentile(v, 95):7.2f}  max {v.max():7.2f}")


if a.plain:
    # The reference's loop, literally: reset(x) from HOST memory, operator(), read_result() -- N times. Under rocprofv3
    # --kernel-trace --stats the trace's average duration of single_kernel is the clock of record; what this prints beside it are the
    # kernel's own stamp (tkspmv_run's return value: first workgroup's entry to the result flag, 100 MHz device clock) and the host
    # clock around the three calls, all from the SAME launches.
    eng = engine()
    own, e2e = [], []
    for i in range(a.plain):
        t0 = time.perf_counter()
        eng.reset(xs[i % 8])
        ns = eng()
        val, idx = eng.read_result()
        e2e.append((time.perf_counter() - t0) * 1e6)
        own.append(ns / 1e3)
    own, e2e = np.array(own[2:]), np.array(e2e[2:])
    print(f"{a.plain} queries through reset / operator() / read_result (2 dropped), {a.rows} x {a.cols}, k = {a.k}:")
    print(f"  device_us_self_stamped (tkspmv_run's return value): median {np.median(own):.2f}  p95 {np.percentile(own, 95):.2f}  min {own.min():.2f}")
    print(f"  end_to_end_us (host clock around the three calls):  median {np.median(e2e):.2f}  p95 {np.percentile(e2e, 95):.2f}  min {e2e.min():.2f}")
    lp, kn = eng.time_host_loop(xs, a.plain)
    print(f"  the same loop in native code (tkspmv_time_host_loop): end to end median {np.median(lp[2:]):.2f}  p95 {np.percentile(lp[2:], 95):.2f}  min {lp[2:].min():.2f}; "
          f"own stamp median {np.median(kn[2:]):.2f}")
    print(f"  counters: {eng.debug_counters()}")
    eng.close()
    sys.exit(0)

eng = engine()
info = eng.info()
alg = info["algorithmic_bytes"]
print("engine:", {k_: info[k_] for k_ in ("grid", "block", "n_wave_partitions", "packets_per_partition", "packed_bytes")})
ns = []
for i in range(42):
    eng.reset_device(dxs[i % 8].data_ptr())
    ns.append(eng())
ns = np.array(ns[2:]) / 1e3
print("single fused launch, hipEvent bracket (us), 40 runs after 2 dropped:")
pct(ns, "tkspmv_run kernel_ns")
t = eng.profile(dxs.data_ptr(), 8, 100)
print(f"  back-to-back batch: {t['query_ns'] / 1e3:.2f} us/query; bracketed single launches back to back: {t['stream_kernel_ns'] / 1e3:.2f} us; "
      f"the bracket around an empty kernel of the same geometry: {t['event_bracket_ns'] / 1e3:.2f} us")
print(f"  => frac of HBM peak at the median bracket: {alg / (np.median(ns) * 1e3) / 8000:.3f}")
# host boundary
eng.reset(xs[0]); eng(); eng.read_result()
t0 = time.perf_counter()
for i in range(300):
    eng.reset(xs[i % 8]); eng(); eng.read_result()
print(f"host boundary (set_query + run + read), per query: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us")
eng.close()

# ---- timeline of single launches ------------------------------------------------------------------------------------
eng = engine(TKSPMV_TRACE="1")
grid = info["grid"]
for i in range(8):
    eng.reset_device(dxs[i % 8].data_ptr())
    eng()
words = 4 * (grid + 1) * 9 * 8
buf = np.zeros(words, dtype=np.uint64)
got = C.c_uint64()
_lib.check(_lib.lib().tkspmv_debug_trace(eng._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), words, C.byref(got)))
tr = buf.reshape(4, grid + 1, 9, 8).astype(np.int64)
names = ["entry", "x staged", "first packet reduced", "loop done", "deferred judged", "flush done"]
for s in range(4):
    w = tr[s, :grid, :8, :]
    live = w[..., 3] > 0
    if not live.any():
        continue
    base = w[..., 0][live].min()
    print(f"single launch, trace slot {s} (us since the first wave's entry; {int(live.sum())} streaming waves):")
    for j, nm in enumerate(names):
        pct((w[..., j][live] - base) * 0.01, nm)
    srv = tr[s, :grid, 8, :]
    ok = srv[:, 5] > 0
    if ok.any():
        pct((srv[ok, 5] - base) * 0.01, "server wave exit")
    last = max(int(w[..., 5][live].max()), int(srv[ok, 5].max()) if ok.any() else 0)
    print(f"  last stamp of the launch: {(last - base) * 0.01:.2f} us")
eng.close()

# ---- selection tail stamps ----------------------------------------------------------------------------------------------
eng = engine(TKSPMV_STAMPS="1")
for i in range(4):
    eng.reset_device(dxs[i % 8].data_ptr())
    eng()
eng.profile(dxs.data_ptr(), 8, 20)  # prints the stamps of the last fused launch to stderr
eng.close()
