#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Top-K SpMV hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one query: one pass of the Top-K SpMV hot path over the whole matrix for a fresh dense vector x, its exact
top-K selected on the device. Queries are issued back to back on one stream (the engine launches its batch kernel once per
32 queries; every query still streams the whole matrix). All inputs are resident in HBM before the timed region; results
stay in HBM. The timed region is bracketed by a barrier + device synchronisation on both sides AND by a hipEvent pair on the
engine's stream (`roofline` uses the events, `value` the host clock).

N = 1 (default): BASELINE.json configs[1] -- 1M x 1024, gamma 20 nnz/row, K = 100, fp32, synthetic (own seeded generator
restating create_matrices.py's distributions). `value` is measured CACHE-DEFEATED: the engine keeps 4 copies of the 118 MB
packet stream and rotates them per query, so no query is served from the 256 MiB Infinity Cache and the roofline fraction
is an honest HBM fraction. Beside the headline the line carries: `parity_checked` (the last timed query against the CPU
oracle), `timing` (median / p95 over >= 30 repetitions, first 2 dropped: the reference's hygiene,
host_spmv_bscsr.cpp:699), `single_query` (the literal reference loop: one reset -> operator() -> read_result at a time),
`configs` (BASELINE configs[2] and configs[4]), `cache_warm`, `multi_query`, `cpu_baseline` (fp64 and fp32).

N > 1, or --total-rows R at any N: BASELINE.json configs[3] -- ONE R-row matrix (default 10M x 1024, gamma 20 nnz/row,
seed 4) cut into N contiguous row shards balanced by nnz; every rank builds only its own shard (same generator seed
everywhere), runs its engine with global row ids, and per query the ranks exchange K (row, score) pairs with ONE RCCL
all-gather (32 queries per exchange, on a side stream) followed by a merge kernel. STRONG scaling: `value` = global queries
per second = steps / max-over-ranks time. The exchange must be the native one (csrc/dist.hip); without it the run fails
loudly unless TKSPMV_DIST=torch asks for the torch.distributed exchange.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--cols", type=int, default=1024)
    ap.add_argument("--nnz", type=int, default=20)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--config3-rows", type=int, default=10000000,
                    help="--gpus N > 1: rows of the BASELINE configs[3] matrix sharded over the same GPUs beside the headline (0 = skip)")
    ap.add_argument("--total-rows", type=int, default=0,
                    help="rows of the ONE matrix cut across the ranks (default: --rows, the matrix of the --gpus 1 line); given at "
                         "--gpus 1 it runs the sharded code path with one shard")
    ap.add_argument("--replicas", type=int, default=4, help="packet-stream copies rotated per query (cache defeat)")
    ap.add_argument("--reps", type=int, default=32, help="repetitions of the timed batch for median / p95 (first 2 dropped)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--queries", type=int, default=64, help="distinct query vectors resident in HBM")
    ap.add_argument("--nnz-per-lane", type=int, default=0, help="entries per lane and packet (0 = the engine's default)")
    ap.add_argument("--waves-per-cu", type=int, default=0, help="streaming waves per CU (0 = the engine's default)")
    ap.add_argument("--threads-per-wg", type=int, default=0, help="streaming threads per workgroup (0 = the engine's default)")
    ap.add_argument("--headline-only", action="store_true",
                    help="profiler runs: nothing but the headline's queries (warm-up, timed region, repetitions) -- no side leg, no nonstationary "
                         "stream, no CPU baseline, no traffic pass; the kernel trace of such a run holds the headline kernel alone")
    ap.add_argument("--skip-warm", action="store_true",
                    help="profiler runs: nothing but the headline queries (no side legs, homogeneous launches for rocprofv3)")
    ap.add_argument("--multi-q", type=int, nargs="*", default=[4, 8],
                    help="queries per matrix pass of the multi-query leg (reported beside the headline; empty = skip)")
    ap.add_argument("--multi-only", type=int, default=0,
                    help="profiler runs: launch nothing but the multi-query path with this many queries per pass")
    ap.add_argument("--traffic", choices=["auto", "live", "static", "off"], default="auto",
                    help="roofline.traffic: a live rocprofv3 --pmc pass over a child run (auto: when rocprofv3 is on PATH), "
                         "or the figure committed under profiles/ (labelled static)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the child of the live traffic pass
    a = ap.parse_args()
    if a.headline_only:
        a.skip_warm = True
    return a


def workload_name(rows, cols, nnz_per_row, nnz, k, replicas):
    """config.workload: the same text for the --gpus 1 line and the --gpus N lines of one matrix (how it is spread over GPUs is
    config.parallelism)."""
    return (f"{rows}x{cols} gamma nnz/row={nnz_per_row} (nnz={nnz}) K={k} fp32, queries back to back, cache-defeated "
            f"({replicas} rotating stream copies)")


def pct(v, p):
    import numpy as np
    return float(np.percentile(np.asarray(v, dtype=np.float64), p))


# ---- CPU baseline --------------------------------------------------------------------------------------
def cpu_baseline(mod, m, xs, k, seconds):
    """The reference's CPU path restated (oracle/oracle.c: sparse_dot_topn's threaded kernel for an N x 1 right-hand side,
    test_cpu.py:99-105) + the global top-k a user does next, timed natively on the host cores: fp64 (what the reference
    runs) and fp32 variants, SpMV-only and SpMV + top-k, median of >= 10 runs after 2 warm-ups. Bounded sample."""
    import numpy as np
    import oracle_lib as O
    hw = os.cpu_count() or 1
    ptr, idx, v = O.coo_to_csr_f64(m.row, m.col, m.val, m.rows)
    t_start = time.perf_counter()
    # thread count: the reference used n_jobs = 40 on 80 hardware threads; here the fastest of a few candidates
    cands = sorted({t for t in (8, 16, 32, 64, 128, hw) if t <= hw}) or [1]
    best_t, best = cands[0], float("inf")
    for t in cands:
        _, tot = O.cpu_bench(ptr, idx, v, m.rows, xs, k, t, 1, 3)
        if float(np.median(tot)) < best:
            best_t, best = t, float(np.median(tot))
    out = {"unit": "queries/s", "cores": best_t, "host_threads": hw, "kind": "port"}
    budget = max(seconds - (time.perf_counter() - t_start), 2.0)
    n_total = 0
    for name, f32 in (("fp64", False), ("fp32", True)):
        reps = int(min(max(10, (budget / 2) / max(best * 1e-3, 1e-4)), 400))
        sp, tot = O.cpu_bench(ptr, idx, v, m.rows, xs, k, best_t, 2, reps, use_f32=f32)
        n_total += reps + 2
        out[name] = {"runs": reps, "spmv_only_ms_median": float(np.median(sp)), "spmv_topk_ms_median": float(np.median(tot)),
                     "spmv_topk_ms_p95": pct(tot, 95), "queries_per_sec": 1e3 / float(np.median(tot))}
    out["value"] = out["fp64"]["queries_per_sec"]
    out["ms_per_query"] = out["fp64"]["spmv_topk_ms_median"]
    out["spmv_only_ms"] = out["fp64"]["spmv_only_ms_median"]
    out["sample"] = (f"{n_total} queries on the same {m.rows}x{m.cols} matrix ({m.nnz} nnz), {best_t} threads (fastest of "
                     f"{cands}): CSR row-block-threaded SpMV (sparse_dot_topn restated, threads created per query as the "
                     f"package does) + global top-{k}; fp64 = the reference's precision (value), fp32 beside it; medians; "
                     f"{time.perf_counter() - t_start:.1f} s of wall time")
    if O.have_ref():  # the reference's own gold (single thread), when oracle/_ref was built in the build container
        reps, t1 = 0, time.perf_counter()
        while reps < 3:
            O.ref_gold_topk(m.row, m.col, m.val, xs[reps % xs.shape[0]], k)
            reps += 1
        out["reference_gold_ms_per_query"] = 1e3 * (time.perf_counter() - t1) / reps
        out["reference_gold_note"] = "spmv_coo_gold_top_k + sort_tuples compiled from the reference headers, 1 thread"
    return out


# ---- parity of a result against the oracle ----------------------------------------------------------------------------
def check_parity(mod, m, x, k, idx, val, eng=None, bit_exact=True):
    """The north-star bar for one query: same index set as the CPU gold (only k-th-boundary near-ties, <= 2e-6 relative in
    fp64, may differ -- counted), scores within 1e-4 relative; and, with the engine at hand, bit for bit against the
    order-matched oracle on the engine's own packing."""
    import numpy as np
    import oracle_lib as O
    out = {"tolerance": "same index set as the gold (k-th-boundary near-ties <= 2e-6 rel. counted), scores within 1e-4 rel."}
    gi, gv = O.gold_topk(m.row, m.col, m.val, x, k)
    swaps, ok = 0, True
    if set(idx.tolist()) != set(gi.tolist()):
        y64, _ = O.scores_f64(m.row, m.col, m.val, x, m.rows)
        kth = np.sort(y64)[-k]
        diff = set(idx.tolist()) ^ set(gi.tolist())
        ok = all(abs(y64[r] - kth) <= 2e-6 * kth for r in diff)
        swaps = len(diff) // 2
    ok = ok and bool(np.allclose(np.sort(val)[::-1], np.sort(gv)[::-1], rtol=1e-4, atol=0)) and bool(np.all(val[:-1] >= val[1:]))
    out.update(vs_gold=ok, boundary_tie_swaps=swaps)
    if eng is not None and bit_exact:
        info = eng.info()
        C = info["packet_entries"] // 64
        packed = mod.Packed(m, k=k, nnz_per_lane=C, n_wave_partitions=(info.get("batch_mode", 0) >> 16) or info["n_wave_partitions"])
        yp, present = O.packed_scores(packed.raw(), x, m.rows, C)
        ei, ev = O.select_topk(yp, present, k, 0.0, info.get("first_row", 0))
        out["bit_exact_vs_order_matched_oracle"] = bool(np.array_equal(idx, ei) and np.array_equal(val.view(np.uint32), ev.view(np.uint32)))
        ok = ok and out["bit_exact_vs_order_matched_oracle"]
    return ok, out


# ---- live HBM traffic (rocprofv3 --pmc over a child run) --------------------------------------------------------------------
def pmc_child(a):
    """What the counter passes profile: the headline engine and nothing but `steps` headline queries."""
    import numpy as np
    import torch
    import _pkg
    mod = _pkg.load()
    m = mod.generate_matrix(a.rows, a.cols, a.nnz, "gamma", 2)
    xs = np.stack([mod.create_sample_vector(a.cols, True, False, True, 1000 + i) for i in range(a.queries)])
    dxs = torch.from_numpy(xs).cuda()
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=0, stream_replicas=a.replicas,
                   nnz_per_lane=a.nnz_per_lane, waves_per_cu=a.waves_per_cu, threads_per_wg=a.threads_per_wg)
    eng.time_queries(dxs.data_ptr(), a.queries, a.steps)
    eng.close()


def live_traffic(a, n_queries=128):
    """HBM bytes per query of the headline kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one
    pass; no trace domain beside --kernel-trace), each over a child run of `n_queries` headline queries. MI355X_MICROARCH.md:
    FETCH_SIZE is tallied in KiB at 64 B per 128-B request on gfx950 (doubled here); WRITE_SIZE reads bytes exactly.
    Returns (bytes_per_query, detail) or (None, reason)."""
    import csv
    import glob
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    got = {}
    tmp = tempfile.mkdtemp(prefix="tkspmv_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-child", "--steps", str(n_queries), "--rows", str(a.rows), "--cols",
                   str(a.cols), "--nnz", str(a.nnz), "--k", str(a.k), "--replicas", str(a.replicas), "--queries", str(a.queries),
                   "--nnz-per-lane", str(a.nnz_per_lane), "--waves-per-cu", str(a.waves_per_cu), "--threads-per-wg",
                   str(a.threads_per_wg)]
            # (TKSPMV_AUTOTUNE=0: the launches tkspmv_create measures its pacing with would be summed into the traffic)
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", TKSPMV_AUTOTUNE="0"), capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"
            total, launches = 0.0, set()
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "batch_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        total += float(row["Counter_Value"])
                        launches.add(row["Dispatch_Id"])
            if not launches:
                return None, f"no batch_kernel rows in the {counter} pass"
            got[counter] = total / n_queries
        fetch = got["FETCH_SIZE"] * 1024.0 * 2.0
        write = got["WRITE_SIZE"] * 1024.0
        return fetch + write, {"fetch_bytes_corrected": fetch, "write_bytes": write, "queries_profiled": n_queries,
                               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over a child run of "
                                         "this workload; FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024, summed "
                                         "over the batch-kernel launches and divided by the queries they served"}
    except Exception as e:  # noqa: BLE001 -- a failed counter pass must never cost the bench line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def static_traffic():
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("stream_kernel_hbm_bytes_per_launch")
    except Exception:  # noqa: BLE001
        return None


# ---- side legs (N = 1) -------------------------------------------------------------------------------------------------------
def multi_query_leg(mod, m, dxs, a, device, alg_bytes):
    """Extension, reported beside the headline: several queries share each pass over the matrix
    (tkspmv_enqueue_multi: the wave-sliced ELL copy, one row per lane). Same synthetic matrix and query vectors, cache-
    defeated rotation; every query still gets its exact top-k. Time = one hipEvent pair around the whole sequence."""
    out = []
    for q in a.multi_q:
        eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=device, stream_replicas=a.replicas, multi_q=q)
        info = eng.info()
        if info["multi_q"] != q:
            eng.close()
            continue
        n = max(a.steps // (8 * q) * (8 * q), 128 * q)  # (at least 128 passes: a handful of passes measures launch overhead)
        eng.time_multi(dxs.data_ptr(), a.queries, n)
        ns = min(eng.time_multi(dxs.data_ptr(), a.queries, n) for _ in range(3))
        out.append({"queries_per_pass": q, "value": 1e9 / ns, "unit": "queries/s", "us_per_query": ns / 1e3,
                    "us_per_pass": ns * q / 1e3, "queries_timed": n,
                    "stream_bytes_per_pass": int(info["multi_bytes"]),
                    "hbm_GBps_of_the_pass": info["multi_bytes"] / (ns * q),
                    "per_query_algorithmic_GBps": alg_bytes / ns})
        eng.close()
    return {"kernel": "tkspmv::multi_kernel<Q> over the wave-sliced ELL copy of the matrix (one row per lane; exact top-k "
                      "per query, scores in the gold's sequential fp32 order); a sequence runs as two independent chains of "
                      "launches on two streams (one launch = one pass of one chain)",
            "note": "a pass is bound by LDS reads and instruction issue, not by HBM: per-query algorithmic GB/s is not an "
                    "HBM figure here and is not compared with the roofline",
            "runs": out}


def _mark(what):
    """TKSPMV_BENCH_TRACE=1: a line on stderr as every leg begins (which leg was running when something went wrong)."""
    if os.environ.get("TKSPMV_BENCH_TRACE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {what}", file=sys.stderr, flush=True)


def single_query_leg(mod, m, xs, dxs, a, device, eng, alg_bytes):
    """The literal loop of the reference's hosts (host_spmv_bscsr.cpp:602-632): ONE query in flight -- reset(vec),
    operator()(), read_result() -- through tkspmv_set_query / tkspmv_run / tkspmv_read on host buffers.
    Two clocks, named as what they are:
      * device_us_self_stamped: what tkspmv_run returns -- the launch's own s_memrealtime span (100 MHz), first workgroup's
        entry to the raising of the result flag. It excludes the dispatch latency before the first wave and the kernel's
        last instructions behind the flag, and is NOT comparable with round 2's hipEvent figures.
      * the figure of record for the kernel is rocprofv3's average duration of the same launches, committed under
        profiles/ (r04_single_query_kernel_stats.csv); `kernels_us.single_query_launch_with_event_bracket` of this line
        is the hipEvent bracket (the reference's hw_exec_time clock: ~6 us around an empty kernel).
    end_to_end_us = host clock around the three calls (hw_full_exec_time + reset). >= 30 runs, first 2 dropped."""
    import numpy as np
    n = max(a.reps, 30) + 2
    kern, e2e = [], []
    for i in range(n):
        x = xs[i % xs.shape[0]]
        t0 = time.perf_counter()
        eng.reset(x)
        ns = eng()
        val, idx = eng.read_result()
        e2e.append((time.perf_counter() - t0) * 1e6)
        kern.append(ns / 1e3)
    ok, par = check_parity(mod, m, xs[(n - 1) % xs.shape[0]], a.k, idx, val, eng)
    kern, e2e = kern[2:], e2e[2:]
    med = float(np.median(kern))
    # the same loop in native code (tkspmv_time_host_loop): what a C++ host like the reference's sees per iteration -- the
    # Python figure above carries three ctypes transitions and the result arrays' allocation on top
    native = None
    if hasattr(eng, "time_host_loop"):
        lp, kn = eng.time_host_loop(xs, n)
        native = {"end_to_end_us": float(np.median(lp[2:])), "end_to_end_us_p95": pct(list(lp[2:]), 95),
                  "device_us_self_stamped": float(np.median(kn[2:])),
                  "note": "tkspmv_set_query + tkspmv_run + tkspmv_read per iteration inside ONE native call, host steady clock per iteration"}
    counters = eng.debug_counters()
    single = counters.get("single_launches", 0) > 0
    return {"kernel": ("tkspmv::single_kernel<7> (one launch per query: workgroup-local thresholds carried from the previous query, "
                       "one record per workgroup, workgroup 0 selects -- it loads every record as the workgroup's flag comes up; a failed "
                       "check repeats the query through tkspmv::stream_kernel)") if single else
                      "tkspmv::stream_kernel<4,false,1024,7,3> (one fused launch per query: stream, flush, in-launch selection)",
            "runs": len(kern), "dropped": 2, "device_us_self_stamped": med, "device_us_self_stamped_p95": pct(kern, 95),
            "kernel_us": med, "kernel_us_p95": pct(kern, 95), "frac": alg_bytes / (med * 1e3) / HBM_PEAK_GBS,
            "end_to_end_us": float(np.median(e2e)), "end_to_end_us_p95": pct(e2e, 95), "parity_checked": ok,
            "end_to_end_clock": "host clock of THIS Python process around reset / __call__ / read_result (ctypes); `native_loop` is the same loop in C",
            "native_loop": native,
            "counters": {k_: counters[k_] for k_ in ("single_launches", "single_repairs", "single_checks_failed") if k_ in counters},
            "clock": "kernel_us = device_us_self_stamped: the launch's own s_memrealtime span at 100 MHz (10 ns ticks), first "
                     "workgroup's entry to the result flag; it excludes dispatch latency and the instructions behind the flag",
            "note": "figure of record for the kernel: rocprofv3 --kernel-trace average of the same launches under profiles/ "
                    "(r04_single_query_kernel_stats.csv); the hipEvent bracket (~6 us around an empty kernel) is "
                    "kernels_us.single_query_launch_with_event_bracket; x travels by CPU stores through the PCIe aperture "
                    "(large BAR), the result by device stores into pinned host memory, verified by a checksum"}


def read_only_floor(eng, info, passes=64):
    """The load-only probe on THIS engine's wave-BSCSR stream (tkspmv_time_stream_read: same launch geometry, rotating copies,
    loads only): what moving that stream costs on this box -- the floor the leg's kernel_us stands against."""
    ns = sorted(eng.time_stream_read(passes) for _ in range(3))[1]
    return {"us": ns / 1e3, "stream_bytes": int(info["packed_bytes"]), "GB_per_s": info["packed_bytes"] / ns if ns else None,
            "method": "tkspmv_time_stream_read on this engine, median of 3 x %d passes" % passes}


def config_legs(mod, a, device):
    """BASELINE.json configs[2] and configs[4] at their own sizes, each with its roofline fraction (and, for the reduced
    precision, the reference's acceptance metric: precision@K against the fp32 gold)."""
    import numpy as np
    import torch
    import oracle_lib as O
    out = []
    # configs[2]: 1M x 1024, 20 nnz/row, K = 8, 32 row partitions (SPMV_PARTITIONS = 32, K lists of 8: host_spmv_bscsr.cpp:133-141)
    m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
    xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(16)])
    dxs = torch.from_numpy(xs).to(torch.device("cuda", device))
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=8, device=device, partitions=32, k_per_partition=8,
                   stream_replicas=a.replicas)
    info = eng.info()
    eng.time_queries(dxs.data_ptr(), 16, 64)
    ns = min(eng.time_queries(dxs.data_ptr(), 16, 512) for _ in range(3))
    val, idx = eng.read_result()
    ok, _ = check_parity(mod, m, xs[511 % 16], 8, idx, val, eng)
    ro = read_only_floor(eng, info)
    out.append({"workload": "configs[2]: 1000000x1024 gamma nnz/row=20, K=8, 32 row partitions x K=8 lists (exact: k <= K per partition)",
                "kernel_us": ns / 1e3, "algorithmic_bytes": int(info["algorithmic_bytes"]),
                "roofline": {"bound": "hbm", "achieved": info["algorithmic_bytes"] / ns, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": info["algorithmic_bytes"] / ns / HBM_PEAK_GBS, "read_only": ro,
                             "kernel_vs_read_only": ro["us"] * 1e3 / ns if ns else None},
                "parity_checked": ok})
    eng.close()
    del eng, m
    # configs[4]: 1M x 512, 40 nnz/row, K = 100, int8 values: Q1.7 bytes (rounded), fp32 x, fp32 accumulate
    m = mod.generate_matrix(1000000, 512, 40, "gamma", 5)
    xs = np.stack([mod.create_sample_vector(512, True, False, True, 1000 + i) for i in range(16)])
    dxs = torch.from_numpy(xs).to(torch.device("cuda", device))
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=device, precision=mod.Q1_7_F32, multi_q=1,
                   stream_replicas=a.replicas)
    info = eng.info()
    eng.time_multi(dxs.data_ptr(), 16, 512)
    runs4 = sorted(eng.time_multi(dxs.data_ptr(), 16, 512) for _ in range(7))
    ns = runs4[3]  # (the median: a run without thresholds -- 50 or 150 us per query, seen while the passes were being built -- must show)
    out_i = torch.zeros(4, a.k, dtype=torch.int32, device=dxs.device)
    out_v = torch.zeros(4, a.k, dtype=torch.float32, device=dxs.device)
    torch.cuda.synchronize()
    eng.enqueue_multi(dxs.data_ptr(), 4, out_i.data_ptr(), out_v.data_ptr())
    eng.synchronize()
    ro4 = read_only_floor(eng, info)
    # (the probe reads the engine's wave-BSCSR byte stream -- 3 B/nnz + row words; the kernel of this leg streams the SELL copy,
    #  2.5 B/nnz: at the probe's rate that stream would take sell_stream_at_that_rate_us)
    ro4["sell_stream_bytes"] = int(info["multi_bytes"])
    ro4["sell_stream_at_that_rate_us"] = info["multi_bytes"] / ro4["GB_per_s"] / 1e3 if ro4["GB_per_s"] else None
    vq = O.round_to_q17(m.val)
    prec, exact = [], True
    for q in range(4):
        gi, _ = O.gold_topk(m.row, m.col, m.val, xs[q], a.k)
        idx = out_i[q].cpu().numpy().view(np.uint32)
        prec.append(len(set(idx.tolist()) & set(gi.tolist())) / a.k)
        y, present = O.scores_f32_segmented(m.row, m.col, vq, xs[q], m.rows)
        ei, ev = O.select_topk(y, present, a.k)
        exact = exact and bool(np.array_equal(idx, ei) and np.array_equal(out_v[q].cpu().numpy().view(np.uint32), ev.view(np.uint32)))
    out.append({"workload": f"configs[4]: 1000000x512 gamma nnz/row=40 (nnz={info['nnz']}), K={a.k}, Q1.7 byte values (rounded to "
                            "nearest), fp32 x, fp32 accumulate; one query per pass over the row-per-lane byte stream, eight passes per launch, "
                            "cache-defeated",
                "dtype": "u8 values / f32 arithmetic", "kernel": "tkspmv::multi_kernel<1,5>",
                "kernel_us": ns / 1e3, "kernel_us_runs": [r / 1e3 for r in runs4], "algorithmic_bytes": int(info["algorithmic_bytes"]),
                "stream_bytes": int(info["multi_bytes"]),
                "roofline": {"bound": "hbm", "achieved": info["algorithmic_bytes"] / ns, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": info["algorithmic_bytes"] / ns / HBM_PEAK_GBS, "read_only": ro4,
                             "physical": {"bytes": int(info["multi_bytes"]), "achieved": info["multi_bytes"] / ns,
                                          "frac": info["multi_bytes"] / ns / HBM_PEAK_GBS}},
                "precision_at_100": float(np.mean(prec)), "precision_at_100_min": float(min(prec)),
                "parity_checked": exact,
                "parity_note": "bit-exact against the order-matched oracle on the de-quantised values (parity unpinned against "
                               "the reference: ap_fixed needs Xilinx headers); precision is against the fp32 gold"})
    eng.close()
    del eng, m
    # configs[3] on ONE GPU: the strong-scaling reference point of `bench.py --gpus N` (the N > 1 lines shard this very
    # matrix; the N = 1 headline above is configs[1], a different workload, so their ratio is not a scaling efficiency)
    try:
        rows3 = 10000000
        m = mod.generate_matrix(rows3, 1024, 20, "gamma", 4)
        xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(16)])
        dxs = torch.from_numpy(xs).to(torch.device("cuda", device))
        eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=device, stream_replicas=2)
        info = eng.info()
        eng.time_queries(dxs.data_ptr(), 16, 32)
        ns = min(eng.time_queries(dxs.data_ptr(), 16, 64) for _ in range(3))
        val, idx = eng.read_result()
        ok, _ = check_parity(mod, m, xs[63 % 16], a.k, idx, val, None, bit_exact=False)
        ro3 = read_only_floor(eng, info, 16)
        out.append({"workload": f"configs[3] on ONE GPU: {rows3}x1024 gamma nnz/row=20 (nnz={info['nnz']}), K={a.k}, fp32, two rotating "
                                "stream copies (1.17 GB each: no cache holds one) -- what `--gpus N` strong-scales",
                    "value": 1e9 / ns, "unit": "queries/s", "kernel_us": ns / 1e3, "algorithmic_bytes": int(info["algorithmic_bytes"]),
                    "roofline": {"bound": "hbm", "achieved": info["algorithmic_bytes"] / ns, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": info["algorithmic_bytes"] / ns / HBM_PEAK_GBS, "read_only": ro3,
                                 "kernel_vs_read_only": ro3["us"] * 1e3 / ns if ns else None},
                    "parity_checked": ok})
        eng.close()
    except Exception as e:  # noqa: BLE001  (a box short of host memory: the leg is reported as missing, the line survives)
        out.append({"workload": "configs[3] on ONE GPU", "error": str(e)})
    return out


def _finish_headline_only(a, info, alg_bytes, kernel_ns, elapsed, reps, read_us, counters, eng):
    """--headline-only: a short line (the run exists for the profiler's trace, not for the record)."""
    eng.close()
    print(json.dumps({"metric": "queries_per_sec", "mode": "headline_only", "value": a.steps / elapsed, "unit": "queries/s", "steps": a.steps,
                      "warmup": a.warmup, "kernel_us": kernel_ns / 1e3, "frac": alg_bytes / kernel_ns / HBM_PEAK_GBS,
                      "sustained_median_us": pct(reps, 50), "p95_over_median": pct(reps, 95) / pct(reps, 50), "read_only_us": read_us,
                      "checks_failed": int(counters["checks_failed"]),
                      "pacing": f"{counters.get('pace_quantum')}x{counters.get('pace_levels')}", "pace_period_ns": counters.get("pace_period_ns"), "pace_tuned_us": counters.get("pace_tuned_us"),
                      "launches_of_the_pacing_measurement": counters.get("pace_tune_launches", 0)}))


# ---- N = 1: BASELINE configs[1] ---------------------------------------------------------------------------------------------
def bench_single(a, mod, torch, np, dev, local_rank):
    m = mod.generate_matrix(a.rows, a.cols, a.nnz, "gamma", 2)
    xs = np.stack([mod.create_sample_vector(a.cols, True, False, True, 1000 + i) for i in range(a.queries)])
    dxs = torch.from_numpy(xs).to(dev)
    if a.multi_only:
        # profiler aid: nothing but multi-query passes (warmup + steps queries), then one JSON line about them
        eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=local_rank, stream_replicas=a.replicas,
                       multi_q=a.multi_only)
        eng.time_multi(dxs.data_ptr(), a.queries, a.warmup)
        ns = eng.time_multi(dxs.data_ptr(), a.queries, a.steps)
        print(json.dumps({"metric": "queries_per_sec", "mode": "multi_only", "queries_per_pass": a.multi_only,
                          "value": 1e9 / ns, "us_per_query": ns / 1e3, "us_per_pass": ns * a.multi_only / 1e3,
                          "steps": a.steps, "warmup": a.warmup}))
        eng.close()
        return
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=local_rank, stream_replicas=a.replicas,
                   nnz_per_lane=a.nnz_per_lane, waves_per_cu=a.waves_per_cu, threads_per_wg=a.threads_per_wg)
    info = eng.info()
    alg_bytes = info["algorithmic_bytes"]
    # ---- warm-up, then EXACTLY `steps` queries between two device synchronisations (+ a hipEvent pair on the engine stream)
    _mark('headline: warm-up + timed region')
    if a.warmup > 0:  # (through the very call the timed region uses: its host path -- ctypes, events -- is warm as well)
        eng.time_queries(dxs.data_ptr(), a.queries, a.warmup)
    # (torch.cuda.synchronize() waits for the whole device, the engine's own stream included: one synchronisation per side)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ns = eng.time_queries(dxs.data_ptr(), a.queries, a.steps)  # enqueue + event pair + wait for the end event
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    host_side = {"timed_call_us": 1e6 * (t1 - t0), "device_us_by_events": kernel_ns * a.steps / 1e3,
                 "synchronisation_after_us": 1e6 * (elapsed - (t1 - t0)),
                 "note": "value uses the host clock over call + synchronisations; roofline the event pair inside the call"}
    # ---- parity of the last timed query against the oracle
    val, idx = eng.read_result()
    parity_ok, parity = check_parity(mod, m, xs[(a.steps - 1) % a.queries], a.k, idx, val, eng)
    # ---- repetitions (reference hygiene: >= 30 runs, first 2 dropped, host_spmv_bscsr.cpp:699). All repetitions are enqueued
    # before the first wait (tkspmv_time_query_batches: one hipEvent between consecutive ones), so the GPU never idles between
    # them: median / p95 of the kernel under sustained load. The same batch timed one call at a time -- each behind a
    # synchronisation and a host-side gap, after which this GPU runs 10-25 % slower for about a millisecond -- is kept beside it.
    _mark('repetitions')
    counters = eng.debug_counters()  # (of the timed region and its warm-up: checks of the local thresholds, repairs, gate)
    n_rep = min(max(a.steps, 32), 512)
    n_reps = max(a.reps, 30) + 2
    # (a GPU that has idled -- the parity check above runs on the host for a second -- streams 10-15 % slower for its first ~20 ms: the
    #  sustained figures are taken behind 96 launches of the same work, tools/launch_series.py)
    eng.time_query_batches(dxs.data_ptr(), a.queries, n_rep, 96)
    reps = [v / 1e3 for v in eng.time_query_batches(dxs.data_ptr(), a.queries, n_rep, n_reps)][2:]
    gap = [eng.time_queries(dxs.data_ptr(), a.queries, n_rep) / 1e3 for _ in range(10)][2:]
    timing = {"repetitions": len(reps), "dropped": 2, "queries_per_repetition": n_rep,
              "kernel_us_median": pct(reps, 50), "kernel_us_p95": pct(reps, 95), "kernel_us_mean": float(np.mean(reps)),
              "p95_over_median": pct(reps, 95) / pct(reps, 50),
              "frac_at_median": alg_bytes / (pct(reps, 50) * 1e3) / HBM_PEAK_GBS,
              "method": "tkspmv_time_query_batches: every repetition enqueued before the first wait, a hipEvent between "
                        "consecutive repetitions (no host-side gap between them)",
              "one_call_per_repetition": {"repetitions": len(gap), "kernel_us_median": pct(gap, 50), "kernel_us_max": max(gap),
                                          "note": "each repetition behind a synchronisation and a host-side gap"}}
    extra = {"parity_checked": parity_ok, "parity": parity, "timing": timing, "host_side": host_side}
    extra["thresholds"] = {"checks_failed": counters["checks_failed"], "carried_thresholds_suspended_for": counters["suspended_for"],
                           "gate_closed_for_launches": counters["local_off_for_launches"], "batch_launches": counters["batch_launches"],
                           "late_repairs": counters.get("late_repairs", 0), "launches_without_repair_launch": counters.get("trusted_launches", 0),
                           "note": "workgroup-local thresholds are checked by every selection; a failed check repeats the query through the "
                                   "exact kernel -- behind the launch in the stream while the host has not seen clean verdicts, from the host's "
                                   "next wait otherwise (counters of the warm-up + timed region)"}
    # ---- what this GPU charges for only LOADING the same stream (engine geometry, no arithmetic): boxes differ by several %
    _mark('load-only floor')
    read_us = sorted(eng.time_stream_read(64) / 1e3 for _ in range(7))[3]
    # (round 5: the same probe on a timetable -- left alone its waves are served unevenly by the memory system, and on some boxes the
    #  unpaced "floor" is a pass the paced product beats: the smallest time over a handful of periods is the floor that means something)
    paced_us, paced_period = read_us, 0
    for rel in (0.90, 0.92, 0.94, 0.96):
        ns_p = int(read_us * 1000 * rel)
        mod.set_option("READ_PROBE_PERIOD", str(ns_p))
        t_p = sorted(eng.time_stream_read(64) / 1e3 for _ in range(3))[1]
        if t_p < paced_us:
            paced_us, paced_period = t_p, ns_p
    mod.set_option("READ_PROBE_PERIOD", None)
    c12 = os.environ.get("TKSPMV_F32_C12", "1") != "0" and a.cols <= 1024 and int(info["packet_entries"]) == 256
    stream_bytes = int(info["n_packets"]) * (1408 if c12 else int(info["packet_entries"]) * 6)  # 12-bit column words: 5.5 B per entry
    read_only = {"us_per_pass": read_us, "stream_bytes": stream_bytes, "GBps": stream_bytes / (read_us * 1e3),
                 "frac_of_peak": stream_bytes / (read_us * 1e3) / HBM_PEAK_GBS,
                 "kernel": "tkspmv::read_probe_kernel<22>: the engine's grid, waves, partitions and non-temporal dwordx4/x2 "
                           "loads, 8 packets in flight per wave, 64 passes in one launch over the rotating stream copies; "
                           "median of 7",
                 "headline_kernel_vs_read_only": read_us / (kernel_ns / 1e3),
                 "median_kernel_vs_read_only": read_us / pct(reps, 50),
                 "paced": {"us_per_pass": paced_us, "period_ns": paced_period, "GBps": stream_bytes / (paced_us * 1e3),
                           "frac_of_peak": stream_bytes / (paced_us * 1e3) / HBM_PEAK_GBS,
                           "note": "the same probe, every wave on a timetable (READ_PROBE_PERIOD; best of 0.90 .. 0.96 of the unpaced time "
                                   "per pass, or the unpaced figure itself): what the memory system gives a kernel that only loads, once "
                                   "its XCDs are served evenly"}}
    extra["exchange_state_bytes"] = int(info.get("state_bytes", 0))
    if not a.skip_warm:
        _mark('single query leg')
        extra["single_query"] = single_query_leg(mod, m, xs, dxs, a, local_rank, eng, alg_bytes)
        _mark('profile()')
        prof = eng.profile(dxs.data_ptr(), a.queries, 100)
        extra["kernels_us"] = {"query_back_to_back": prof["query_ns"] / 1e3,
                               "single_query_launch_with_event_bracket": prof["stream_kernel_ns"] / 1e3,
                               "event_bracket_around_an_empty_kernel": prof["event_bracket_ns"] / 1e3,
                               "spmv_only_variant": prof["scores_kernel_ns"] / 1e3}
        # same matrix every query (fits the Infinity Cache): the steady state of a deployed single-matrix service
        _mark('cache-warm engine')
        warm = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=local_rank)
        warm.enqueue_many(dxs.data_ptr(), a.queries, max(a.warmup, 32))
        warm.synchronize()
        t1 = time.perf_counter()
        warm_ns = warm.time_queries(dxs.data_ptr(), a.queries, max(a.steps, 64))
        warm_elapsed = time.perf_counter() - t1
        warm_ns = min(warm_ns, min(warm.time_query_batches(dxs.data_ptr(), a.queries, 256, 6)[2:]))
        warm_read = sorted(warm.time_stream_read(64) / 1e3 for _ in range(5))[2]
        warm.close()
        extra["cache_warm"] = {"value": max(a.steps, 64) / warm_elapsed, "unit": "queries/s", "kernel_us": warm_ns / 1e3,
                               "achieved_GBps": alg_bytes / warm_ns,
                               # the load-only probe over the SAME single copy (it fits the Infinity Cache): if it sits where the
                               # cache-defeated floor sits, what bounds a streaming kernel here is not HBM but what the CUs can take in
                               "read_only_us_per_pass": warm_read, "kernel_vs_read_only": warm_read / (warm_ns / 1e3),
                               "note": "one matrix (118 MB packed) re-read every query: served largely by the 256 MiB "
                                       "Infinity Cache, not comparable with the HBM roofline"}
        if a.multi_q:
            _mark('multi-query leg')
            extra["multi_query"] = multi_query_leg(mod, m, dxs, a, local_rank, alg_bytes)
    # ---- a stream of queries that is NOT stationary: the same matrix, every query scaled by 1, 0.01 or 3 (drawn per query):
    # carried thresholds are invalidated again and again, checks fail, queries are repeated -- what the mode costs then. (The last
    # leg on this engine: every failed check suspends carried thresholds for 16 .. 4096 further selections, which would colour
    # whatever is measured behind it.)
    if a.headline_only:
        return _finish_headline_only(a, info, alg_bytes, kernel_ns, elapsed, reps, read_us, counters, eng)
    _mark('nonstationary leg')
    rng = np.random.default_rng(11)
    sc = rng.choice([1.0, 0.01, 3.0], size=a.queries).astype(np.float32)
    # (round 4: carried thresholds are relative to the query's L1 norm, so scales alone no longer invalidate them; what does is a
    #  change of the query's DIRECTION: every 16th query here has its sign flipped, every 16th is concentrated on 32 columns)
    xs_ns = xs * sc[:, None]
    c_sc0 = eng.debug_counters()
    dxs_sc = torch.from_numpy(np.ascontiguousarray(xs_ns)).to(dev)  # (scales only, first: its failures -- none expected -- would suspend carrying)
    eng.time_queries(dxs_sc.data_ptr(), a.queries, 64)
    sc_ns = [v / 1e3 for v in eng.time_query_batches(dxs_sc.data_ptr(), a.queries, n_rep, 8)][2:]
    c_sc1 = eng.debug_counters()
    xs_ns[5::16] *= np.float32(-1.0)
    for j in range(11, xs_ns.shape[0], 16):
        keep = rng.choice(xs_ns.shape[1], size=32, replace=False)
        mask = np.zeros(xs_ns.shape[1], dtype=bool)
        mask[keep] = True
        xs_ns[j, ~mask] = 0.0
    dxs_ns = torch.from_numpy(np.ascontiguousarray(xs_ns)).to(dev)
    c0 = eng.debug_counters()
    eng.time_queries(dxs_ns.data_ptr(), a.queries, 64)
    ns_ns = [v / 1e3 for v in eng.time_query_batches(dxs_ns.data_ptr(), a.queries, n_rep, 8)][2:]
    val_ns, idx_ns = eng.read_result()
    ok_ns, _ = check_parity(mod, m, xs_ns[(n_rep - 1) % a.queries], a.k, idx_ns, val_ns, eng)
    c1 = eng.debug_counters()
    extra["nonstationary"] = {"kernel_us_median": pct(ns_ns, 50), "queries": 64 + 8 * n_rep, "scales": [1.0, 0.01, 3.0],
                              "checks_failed": c1["checks_failed"] - c0["checks_failed"],
                              "gate_closed_for_launches": c1["local_off_for_launches"], "parity_checked": ok_ns,
                              "carried_thresholds_suspended_for_after": c1["suspended_for"],
                              "direction_changes": "every 16th query sign-flipped, every 16th concentrated on 32 of the columns",
                              "scales_only": {"kernel_us_median": pct(sc_ns, 50), "checks_failed": c_sc1["checks_failed"] - c_sc0["checks_failed"]},
                              "note": "same matrix, every query scaled by 1, 0.01 or 3 at random (harmless: thresholds are carried relative to "
                                      "the query's L1 norm) and one query in eight changed in direction (which is not): a failed check costs "
                                      "the query a second pass; a launch of which a quarter fails closes the gate for 8+ launches"}
    eng.close()
    if not a.skip_warm:
        # the same workload with 16-bit column words (TKSPMV_F32_C12=0: 6 instead of 5.5 bytes per nnz, same bits)
        _mark('16-bit column words leg')
        os.environ["TKSPMV_F32_C12"] = "0"
        try:
            e16 = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=local_rank, stream_replicas=a.replicas)
            e16.time_queries(dxs.data_ptr(), a.queries, 64)
            k16 = sorted(e16.time_queries(dxs.data_ptr(), a.queries, 512) for _ in range(5))[2]
            r16 = sorted(e16.time_stream_read(64) for _ in range(5))[2]
            extra["f32_c16"] = {"layout": "fp32 values + 16-bit column words (1536-byte packets instead of 1408), TKSPMV_F32_C12=0",
                                "stream_bytes": int(e16.info()["n_packets"]) * 1536, "kernel_us": k16 / 1e3,
                                "read_only_us_per_pass": r16 / 1e3, "frac_of_algorithmic_peak": alg_bytes / k16 / HBM_PEAK_GBS,
                                "note": "8.3 % fewer bytes move the load-only floor by 8 % and the kernel by 2-3 %: the batch kernel "
                                        "is bound by bytes and by entries per second at once (DESIGN.md section 3)"}
            e16.close()
        except Exception as e:  # noqa: BLE001
            extra["f32_c16"] = {"error": str(e)}
        del os.environ["TKSPMV_F32_C12"]
        _mark('configs legs')
        extra["configs"] = config_legs(mod, a, local_rank)
    # ---- HBM traffic of the headline kernel
    traffic, source, detail = None, "off", None
    if a.traffic in ("auto", "live") and not a.skip_warm:
        _mark('live traffic (rocprofv3 --pmc child)')
        traffic, detail = live_traffic(a)
        source = "live" if traffic is not None else "static"
        if traffic is None:
            detail = {"live_pass": detail}
    if traffic is None and a.traffic != "off":
        traffic, source = static_traffic(), "static"
    line = {
        "metric": "queries_per_sec", "value": a.steps / elapsed, "unit": "queries/s", "n_gpus": 1,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
        # (strong: --gpus N cuts THIS matrix into N row shards -- bench_sharded -- so the total work is fixed as N grows)
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_name(a.rows, a.cols, a.nnz, int(info["nnz"]), a.k, a.replicas),
                   "rows": a.rows, "cols": a.cols, "nnz": int(info["nnz"]), "k": a.k, "parallelism": "single GPU",
                   "launch": {"grid": info["grid"], "block": info["block"] + 64,
                              "wave_partitions": info["n_wave_partitions"], "packet_entries": info["packet_entries"],
                              # (tkspmv_info.batch_mode: how the engine runs back-to-back queries on this matrix)
                              "selector_workgroups": info.get("batch_mode", 0) & 0xFF,
                              "thresholds": {0: "device-wide exchange", 1: "workgroup-local (best packet maximum per wave), checked by the selection; an exact launch repeats what failed",
                                             2: "workgroup-local (second best packet maximum per wave), checked by the selection; an exact launch repeats what failed"}[(info.get("batch_mode", 0) >> 8) & 0xFF]}},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / kernel_ns, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / kernel_ns / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source,
                     "traffic_detail": detail,
                     "kernel": "tkspmv::batch_kernel<4,1024,7,false,true> (fp32, 12-bit column words, workgroup-local thresholds; up to 32 "
                               "queries per launch; figures are per query; a failed check is repaired by an exact launch -- in the stream while "
                               "the host has not seen clean verdicts, from the host's next wait otherwise: `late_repairs`)",
                     "algorithmic_bytes": int(alg_bytes), "kernel_us": kernel_ns / 1e3,
                     # scalar keys (nested objects do not survive into the driver's record): the load-only floor of this
                     # box, the kernel against it, the kernel under sustained load, the checks of the timed region and its warm-up
                     "read_only_us": read_us, "kernel_vs_read_only": read_us / (kernel_ns / 1e3),
                     "read_only_paced_us": paced_us, "kernel_vs_read_only_paced": paced_us / (kernel_ns / 1e3),
                     "sustained_median_us": pct(reps, 50), "sustained_p95_us": pct(reps, 95),
                     "p95_over_median": pct(reps, 95) / pct(reps, 50),
                     "sustained_frac": alg_bytes / (pct(reps, 50) * 1e3) / HBM_PEAK_GBS,
                     "sustained_vs_read_only": read_us / pct(reps, 50),
                     "checks_failed": int(counters["checks_failed"]), "late_repairs": int(counters.get("late_repairs", 0)),
                     "launches_without_repair_launch": int(counters.get("trusted_launches", 0)),
                     "pace_period_ns": counters.get("pace_period_ns"),
                     "pacing": (f"timetable of {counters.get('pace_period_ns')} ns per query, pauses by rank {counters.get('pace_quantum')}x{counters.get('pace_levels')} behind it" if counters.get("pace_period_ns")
                                else f"{counters.get('pace_quantum')}x{counters.get('pace_levels')}") + (f" (measured at create, {counters.get('pace_tuned_us')} us)" if counters.get("pace_tuned_us") else " (static)"),
                     "read_only": read_only,
                     # what the memory system physically moves (the stream is 5.5 B/nnz, the algorithmic figure counts 6): the
                     # measured traffic, or the stream's size, over the same kernel time -- `frac` above is the SURVEY 8(d) figure
                     "physical": {"bytes": float(traffic) if traffic else float(stream_bytes),
                                  "source": "measured traffic" if traffic else "stream bytes",
                                  "GBps": (float(traffic) if traffic else float(stream_bytes)) / kernel_ns,
                                  "frac_of_peak": (float(traffic) if traffic else float(stream_bytes)) / kernel_ns / HBM_PEAK_GBS},
                     "method": "one hipEvent pair on the engine stream around the timed region's back-to-back launches, "
                               "duration = event time / steps. A launch of the batch kernel streams the matrix once per "
                               "query for up to 32 queries (one continuous prefetch pipeline per wave) and selects each "
                               "query's top-k in its selector workgroup; rocprofv3's average launch duration / queries per "
                               "launch agrees (profiles/README.md). `timing` repeats the measurement (median, p95)."},
    }
    line.update(extra)
    if a.cpu_seconds > 0:
        line["cpu_baseline"] = cpu_baseline(mod, m, xs, a.k, a.cpu_seconds)
    print(json.dumps(line))


# ---- row-sharded: BASELINE configs[3] ---------------------------------------------------------------------------------------
def bench_sharded(a, mod, torch, np, dev, local_rank, rank, world):
    """--gpus N strong-scales THE workload of the --gpus 1 line: the same 1M x 1024 matrix (seed 2; every row has its own PRNG
    streams, so a rank generates exactly its rows of it), cut into N contiguous row shards balanced by nnz. BASELINE configs[3]
    -- the 10M-row matrix over the same N GPUs -- rides along as the line's `config3` key (--config3-rows 0 skips it)."""
    line = sharded_leg(a, mod, torch, np, dev, local_rank, rank, world, a.total_rows or a.rows, 2, a.steps, a.warmup)
    if a.config3_rows > 0 and (world > 1 or a.total_rows == 0):
        c3 = sharded_leg(a, mod, torch, np, dev, local_rank, rank, world, a.config3_rows, 4, min(a.steps, 256), min(a.warmup, 32))
        if rank == 0:
            line["config3"] = {k: c3[k] for k in ("value", "unit", "ms_per_step", "steps", "config", "roofline", "per_rank", "parity_checked")}
    if rank == 0:
        print(json.dumps(line))


def sharded_leg(a, mod, torch, np, dev, local_rank, rank, world, total_rows, seed, steps, warmup):
    from importlib import import_module
    dist = import_module("torch.distributed")
    dmod = import_module("approximate_spmv_topk_amd.distributed")
    shard, (r0, r1), total_nnz = dmod.generate_shard(total_rows, a.cols, a.nnz, "gamma", seed, rank, world)
    xs = np.stack([mod.create_sample_vector(a.cols, True, False, True, 1000 + i) for i in range(a.queries)])
    dxs = torch.from_numpy(xs).to(dev)
    eng = mod.SpMV(shard.row, shard.col, shard.val, shard.rows, shard.cols, k=a.k, device=local_rank, first_row=r0,
                   stream_replicas=a.replicas, nnz_per_lane=a.nnz_per_lane, waves_per_cu=a.waves_per_cu,
                   threads_per_wg=a.threads_per_wg)
    info = eng.info()
    multi = dist.is_initialized()

    def sync_all():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    use_torch = os.environ.get("TKSPMV_DIST", "native") == "torch"
    native = None
    if not use_torch:
        # The native exchange (RCCL all-gather + merge kernel, csrc/dist.hip) or nothing: a scaling run must not quietly
        # measure a different exchange. TKSPMV_DIST=torch asks for the torch.distributed exchange explicitly.
        try:
            native = dmod.NativeShardedSpMV(eng, dev)
        except Exception as e:  # noqa: BLE001
            print(f"[rank {rank}] the native RCCL exchange is unavailable: {e}\n"
                  f"          (set TKSPMV_DIST=torch to measure the torch.distributed exchange instead)", file=sys.stderr)
            sys.exit(3)
    cross = None
    if native is not None and (multi or os.environ.get("TKSPMV_BENCH_CROSS")):  # (the variable: rehearsal on one GPU)
        # The native exchange has to prove itself before it is timed (it cannot be rehearsed on a one-GPU box: RCCL refuses two
        # ranks on one device): a few queries through it, the last one's merged list against the same query through the
        # torch.distributed exchange. A mismatch, an error or no answer within 3 minutes ends the run with a message and a
        # non-zero exit code -- never a silent switch to the other exchange.
        import signal

        def stuck(signum, frame):  # noqa: ARG001
            print(f"[rank {rank}] the native RCCL exchange did not answer within 180 s (TKSPMV_DIST=torch measures the "
                  "torch.distributed exchange instead)", file=sys.stderr, flush=True)
            os._exit(4)

        signal.signal(signal.SIGALRM, stuck)
        signal.alarm(180)
        n_chk = min(5, a.queries)
        native.run_many(dxs.data_ptr(), a.queries, n_chk)
        native.synchronize()
        nv, ni = native.read()
        sh = dmod.ShardedTopK(a.k, dev)
        idx_v, val_v = sh.local_views()
        torch.cuda.synchronize()
        ts = torch.cuda.Stream(device=dev)  # (a stream of its own: handle 0, torch's default stream, means "the engine's stream")
        with torch.cuda.stream(ts):
            eng.enqueue(dxs[n_chk - 1].data_ptr(), idx_v.data_ptr(), val_v.data_ptr(), ts.cuda_stream)
            ti, tv = sh.step()
        torch.cuda.synchronize()
        signal.alarm(0)
        ti, tv = ti.cpu().numpy().astype(np.uint32), tv.cpu().numpy()
        same = bool(np.array_equal(ni, ti) and np.array_equal(nv.view(np.uint32), tv.view(np.uint32)))
        cross = {"native_vs_torch_exchange_same_list": same, "queries": n_chk}
        if not same and os.environ.get("TKSPMV_BENCH_CROSS"):
            print("native", ni[:6], nv[:6], "torch", ti[:6], tv[:6], file=sys.stderr)
        if not same:
            print(f"[rank {rank}] the native exchange and the torch.distributed exchange disagree on query {n_chk - 1}: "
                  f"{int((ni != ti).sum())} of {a.k} row ids differ", file=sys.stderr, flush=True)
            sys.exit(5)
    if native is not None:
        native.run_many(dxs.data_ptr(), a.queries, warmup)
        native.synchronize()
        sync_all()
        t0 = time.perf_counter()
        native.run_many(dxs.data_ptr(), a.queries, steps)
        native.synchronize()
        sync_all()
        elapsed = time.perf_counter() - t0
        val, idx = native.read()  # merged (global) top-k of the last timed query
        exchange_ns = native.time_exchange(20)
        exchange = {"kind": "native: RCCL ncclAllGather of 2*K int32 per rank and query + merge kernel, 32 queries per exchange, on a "
                            "side stream overlapping the next batch's local kernels (csrc/dist.hip)",
                    "us_per_exchange_batch": exchange_ns / 1e3, "queries_per_exchange": 32,
                    "note": "the exchange alone, back to back (all-gather + merge launch); inside the step it overlaps the local kernels",
                    "cross_check": cross}
        native.close()
    else:
        sh = dmod.ShardedTopK(a.k, dev)
        idx_v, val_v = sh.local_views()
        torch.cuda.synchronize()
        ts = torch.cuda.Stream(device=dev)  # (handle 0 -- torch's default stream -- would mean "the engine's own stream")

        def step(i):
            eng.enqueue(dxs[i % a.queries].data_ptr(), idx_v.data_ptr(), val_v.data_ptr(), ts.cuda_stream)
            return sh.step()

        with torch.cuda.stream(ts):
            for i in range(warmup):
                step(i)
        sync_all()
        t0 = time.perf_counter()
        with torch.cuda.stream(ts):
            for i in range(steps):
                ei, ev = step(i)
        sync_all()
        elapsed = time.perf_counter() - t0
        idx, val = ei.cpu().numpy().astype(np.uint32), ev.cpu().numpy()
        exchange = {"kind": "torch.distributed all_gather_into_tensor of 2*K int32 per rank + on-device merge, every step "
                            "(TKSPMV_DIST=torch)"}
    kernel_ns = eng.time_queries(dxs.data_ptr(), a.queries, min(max(steps, 64), 512))  # this rank's local kernel alone
    read_ns = sorted(eng.time_stream_read(64) for _ in range(3))[1]  # this shard's load-only floor on this rank's GPU
    per_rank = [{"rank": rank, "rows": r1 - r0, "first_row": r0, "nnz": int(info["nnz"]), "kernel_us": kernel_ns / 1e3,
                 "algorithmic_bytes": int(info["algorithmic_bytes"]),
                 "frac": info["algorithmic_bytes"] / kernel_ns / HBM_PEAK_GBS,
                 "read_only_us": read_ns / 1e3, "batch_mode": {"selector_workgroups": info.get("batch_mode", 0) & 0xFF,
                                                              "local_thresholds": (info.get("batch_mode", 0) >> 8) & 0xFF}}]
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    parity_ok, parity, line = None, None, None
    if rank == 0:
        # the merged result of the last timed query against the gold over the WHOLE matrix (built here, outside the timed
        # region, on rank 0 only)
        whole = mod.generate_matrix(total_rows, a.cols, a.nnz, "gamma", seed)
        assert whole.nnz == total_nnz
        parity_ok, parity = check_parity(mod, whole, xs[(steps - 1) % a.queries], a.k, idx, val, None, bit_exact=False)
        del whole
        slowest = max(per_rank, key=lambda r: r["kernel_us"])
        line = {
            "metric": "queries_per_sec", "value": steps / elapsed, "unit": "queries/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name(total_rows, a.cols, a.nnz, total_nnz, a.k, a.replicas),
                       "rows": total_rows, "cols": a.cols, "nnz": total_nnz, "k": a.k, "n_shards": world,
                       "parallelism": f"row-shard x{world}: ONE matrix cut into {world} contiguous row shards balanced by nnz; per "
                                      f"query one RCCL all-gather of K (row, score) pairs per rank + merge"},
            "roofline": {"bound": "hbm", "achieved": slowest["algorithmic_bytes"] / (slowest["kernel_us"] * 1e3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": slowest["frac"], "traffic": None,
                         "kernel": "tkspmv::batch_kernel<4,1024,7> on the slowest rank's shard (per query)",
                         "algorithmic_bytes": slowest["algorithmic_bytes"], "kernel_us": slowest["kernel_us"],
                         "method": "per rank: one hipEvent pair around a batch of back-to-back local launches, no exchange"},
            "per_rank": per_rank, "exchange": exchange, "parity_checked": parity_ok, "parity": parity,
        }
    eng.close()
    return line if rank == 0 else None


def main():
    a = parse()
    if a.pmc_child:
        pmc_child(a)
        return
    import numpy as np
    import torch
    import _pkg
    mod = _pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    # (TKSPMV_BENCH_DEVICE / TKSPMV_BENCH_PG: rehearsal aids -- several ranks on ONE GPU, process group over gloo)
    local_rank = int(os.environ.get("TKSPMV_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world == 1 and a.gpus > 1:
        print("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the Top-K SpMV engine has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # TKSPMV_BENCH_FORCE_DIST=1 under torch.distributed.run with ONE rank takes the process-group code path (a one-rank
    # RCCL communicator): the way to exercise that path on a single-GPU box.
    multi = world > 1 or (os.environ.get("TKSPMV_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if multi:
        import torch.distributed as dist
        if os.environ.get("TKSPMV_BENCH_PG") == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    try:
        if multi or a.total_rows:
            bench_sharded(a, mod, torch, np, dev, local_rank, rank, world)
        else:
            bench_single(a, mod, torch, np, dev, local_rank)
    finally:
        if multi:
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
