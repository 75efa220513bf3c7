#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Top-K SpMV hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one query: one pass of the Top-K SpMV hot path over the whole matrix for a fresh dense vector x. The
queries are issued back to back on one stream (tkspmv_enqueue_many: the engine launches its batch kernel once per 32
queries; every query still streams the whole matrix and is selected exactly).
Workload at N=1 = BASELINE.json configs[1]: 1M x 1024, 20 nnz/row (gamma), K=100, fp32, synthetic (own seeded
generator restating create_matrices.py's distributions). All inputs (packed matrix, 64 query vectors) are
resident in HBM before the timed region; results stay in HBM.

`value` is measured in CACHE-DEFEATED mode: the engine keeps 4 copies of the 118 MB packet stream and rotates
them per query, so no query can be served from the 256 MiB Infinity Cache and the roofline fraction is an honest
HBM fraction. The steady-state number for ONE matrix (which fits the Infinity Cache) is reported as `cache_warm`.

N > 1 (weak scaling): every rank owns a 1M-row shard of an (N x 1M)-row matrix; per step each rank runs its local
engine, then ONE RCCL all-gather of K (row, score) pairs per rank and the merge (32 steps per exchange, pipelined). `value` = N * steps / time
(1M-row-shard queries per second, whole job); `global_queries_per_sec` = steps / time.
"""
import argparse
import json
import os
import sys
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
    ap.add_argument("--replicas", type=int, default=4, help="packet-stream copies rotated per query (cache defeat)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--queries", type=int, default=64, help="distinct query vectors resident in HBM")
    ap.add_argument("--nnz-per-lane", type=int, default=0, help="entries per lane and packet (0 = the engine's default)")
    ap.add_argument("--waves-per-cu", type=int, default=0, help="streaming waves per CU (0 = the engine's default)")
    ap.add_argument("--threads-per-wg", type=int, default=0, help="streaming threads per workgroup (0 = the engine's default)")
    ap.add_argument("--skip-warm", action="store_true", help="skip the cache-warm leg (homogeneous launches for rocprofv3)")
    ap.add_argument("--multi-q", type=int, nargs="*", default=[4, 8],
                    help="queries per matrix pass of the multi-query leg (reported beside the headline; empty = skip)")
    ap.add_argument("--multi-only", type=int, default=0,
                    help="profiler runs: launch nothing but the multi-query path with this many queries per pass")
    return ap.parse_args()


def cpu_baseline(mod, m, xs, k, seconds):
    """The reference's CPU path restated (oracle/oracle.c: sparse_dot_topn's threaded kernel for an N x 1
    right-hand side, fp64, n_jobs = all host threads) + the global top-k a user does next. Bounded sample."""
    import numpy as np
    import oracle_lib as O
    cores = os.cpu_count() or 1
    ptr, idx, v = O.coo_to_csr_f64(m.row, m.col, m.val, m.rows)
    n = 0
    t_spmv = 0.0
    t0 = time.perf_counter()
    while True:
        x = xs[n % xs.shape[0]].astype(np.float64)
        a = time.perf_counter()
        scores, kept = O.cpu_topn(ptr, idx, v, m.rows, x, 0.0, cores)
        b = time.perf_counter()
        O.cpu_global_topk(scores, kept, k)
        t_spmv += b - a
        n += 1
        if time.perf_counter() - t0 >= seconds and n >= 3:
            break
    total = time.perf_counter() - t0
    out = {"value": n / total, "unit": "queries/s", "cores": cores, "kind": "port",
           "sample": f"{n} queries on the same {m.rows}x{m.cols} matrix ({m.nnz} nnz): fp64 CSR row-block-threaded "
                     f"SpMV (sparse_dot_topn restated) + global top-{k}; {total:.1f} s of wall time",
           "spmv_only_ms": 1e3 * t_spmv / n, "ms_per_query": 1e3 * total / n}
    # the reference's own gold (single thread), when oracle/_ref was built in the build container
    if O.have_ref():
        reps, t1 = 0, time.perf_counter()
        while reps < 3:
            O.ref_gold_topk(m.row, m.col, m.val, xs[reps % xs.shape[0]], k)
            reps += 1
        out["reference_gold_ms_per_query"] = 1e3 * (time.perf_counter() - t1) / reps
        out["reference_gold_note"] = "spmv_coo_gold_top_k + sort_tuples compiled from the reference headers, 1 thread"
    return out


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
        n = max(a.steps // (8 * q) * (8 * q), 8 * q)
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


def main():
    a = parse()
    import numpy as np
    import torch
    import _pkg
    mod = _pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world == 1 and a.gpus > 1:
        print("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the Top-K SpMV engine has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # TKSPMV_BENCH_FORCE_DIST=1 under torch.distributed.run with ONE rank takes the N > 1 code path (a one-rank
    # RCCL communicator): the way to exercise that path on a single-GPU box.
    multi = world > 1 or (os.environ.get("TKSPMV_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if multi:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    # ---- workload: rank r owns shard r (own seed); N=1 is BASELINE configs[1] -----------------------------------
    m = mod.generate_matrix(a.rows, a.cols, a.nnz, "gamma", 2 + rank)
    xs = np.stack([mod.create_sample_vector(a.cols, True, False, True, 1000 + i) for i in range(a.queries)])
    dxs = torch.from_numpy(xs).to(dev)
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=local_rank, first_row=rank * a.rows,
                   stream_replicas=a.replicas, nnz_per_lane=a.nnz_per_lane, waves_per_cu=a.waves_per_cu,
                   threads_per_wg=a.threads_per_wg)
    info = eng.info()
    alg_bytes = info["algorithmic_bytes"]

    def sync_all():
        torch.cuda.synchronize()
        if multi:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    if not multi and a.multi_only:
        # profiler aid: nothing but multi-query passes (warmup + steps queries), then one JSON line about them
        eng.close()
        eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=local_rank, stream_replicas=a.replicas,
                       multi_q=a.multi_only)
        eng.time_multi(dxs.data_ptr(), a.queries, a.warmup)
        ns = eng.time_multi(dxs.data_ptr(), a.queries, a.steps)
        print(json.dumps({"metric": "queries_per_sec", "mode": "multi_only", "queries_per_pass": a.multi_only,
                          "value": 1e9 / ns, "us_per_query": ns / 1e3, "us_per_pass": ns * a.multi_only / 1e3,
                          "steps": a.steps, "warmup": a.warmup}))
        eng.close()
        return
    if not multi:
        # ---- N = 1: K queries back to back on the engine stream -------------------------------------------------
        eng.enqueue_many(dxs.data_ptr(), a.queries, a.warmup)
        eng.synchronize()
        sync_all()
        t0 = time.perf_counter()
        eng.enqueue_many(dxs.data_ptr(), a.queries, a.steps)
        eng.synchronize()
        sync_all()
        elapsed = time.perf_counter() - t0
        # kernel-level timing (hipEvents on the engine stream), same rotation. --skip-warm (profiler runs) launches
        # nothing but the queries themselves, so the rocprofv3 per-kernel average is that of the timed launches.
        n_prof = min(max(a.steps, 50), 500)
        if a.skip_warm:
            prof = {"query_ns": eng.time_queries(dxs.data_ptr(), a.queries, n_prof)}
        else:
            prof = eng.profile(dxs.data_ptr(), a.queries, n_prof)
        # sanity: the last query's result against the oracle order of scores
        val, idx = eng.read_result()
        assert np.all(val[:-1] >= val[1:]) and len(set(idx.tolist())) == a.k
        # same matrix every query (fits the Infinity Cache): the steady state of a deployed single-matrix service
        cache_warm = None
        if not a.skip_warm:
            warm = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=local_rank)
            warm.enqueue_many(dxs.data_ptr(), a.queries, a.warmup)
            warm.synchronize()
            t1 = time.perf_counter()
            warm.enqueue_many(dxs.data_ptr(), a.queries, a.steps)
            warm.synchronize()
            warm_elapsed = time.perf_counter() - t1
            warm_ns = warm.time_queries(dxs.data_ptr(), a.queries, n_prof)
            warm.close()
            cache_warm = {"value": a.steps / warm_elapsed, "unit": "queries/s",
                          "ms_per_step": 1e3 * warm_elapsed / a.steps,
                          "kernel_us": warm_ns / 1e3,
                          "achieved_GBps": alg_bytes / warm_ns,
                          "note": "one matrix (118 MB packed) re-read every query: served largely by the 256 MiB "
                                  "Infinity Cache, not comparable with the HBM roofline"}
        units = a.steps
        extra = {"cache_warm": cache_warm}
        if not a.skip_warm and a.multi_q:
            extra["multi_query"] = multi_query_leg(mod, m, dxs, a, local_rank, alg_bytes)
        if "stream_kernel_ns" in prof:
            extra["kernels_us"] = {"query_back_to_back": prof["query_ns"] / 1e3,
                                   "single_query_fused_launch_with_event_bracket": prof["stream_kernel_ns"] / 1e3,
                                   "spmv_only_variant": prof["scores_kernel_ns"] / 1e3}
        kernel_ns = prof["query_ns"]  # launches back to back, no gaps => batch time / queries = kernel time per query
    else:
        # ---- N > 1: local engine -> all-gather of K pairs -> merge, per step ---------------------------------------
        import torch.distributed as dist
        from importlib import import_module
        dmod = import_module("approximate_spmv_topk_amd.distributed")
        native = None
        if os.environ.get("TKSPMV_DIST", "native") == "native":
            try:  # native RCCL exchange (csrc/dist.hip); any rank failing makes every rank fall back
                native = dmod.NativeShardedSpMV(eng, dev)
                ok = torch.ones(1, device=dev)
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] native exchange unavailable ({e}); using torch.distributed", file=sys.stderr)
                ok = torch.zeros(1, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() == 0 and native is not None:
                native.close()
                native = None
        if native is not None:
            native.run_many(dxs.data_ptr(), a.queries, a.warmup)
            native.synchronize()
            sync_all()
            t0 = time.perf_counter()
            native.run_many(dxs.data_ptr(), a.queries, a.steps)
            native.synchronize()
            sync_all()
            elapsed = time.perf_counter() - t0
            exchange = ("native: RCCL ncclAllGather of 2*K int32 per rank and query + merge kernel, 32 queries per exchange, on "
                        "a side stream overlapping the next batch's local kernel (csrc/dist.hip)")
            # outside the timed region: the last query again through torch.distributed's all-gather + torch merge
            val_n, idx_n = native.read()
            native.close()
            sh = dmod.ShardedTopK(a.k, dev)
            idx_v, val_v = sh.local_views()
            eng.enqueue(dxs[(a.steps - 1) % a.queries].data_ptr(), idx_v.data_ptr(), val_v.data_ptr(),
                        torch.cuda.current_stream().cuda_stream)
            ei, ev = sh.step()
            same = (np.array_equal(ei.cpu().numpy().astype(np.uint32), idx_n)
                    and np.array_equal(ev.cpu().numpy(), val_n))
            exchange += "; cross-check against the torch.distributed exchange: " + ("identical" if same else "MISMATCH")
            if not same:
                print(f"[rank {rank}] native exchange result differs from the torch.distributed exchange", file=sys.stderr)
        else:
            sh = dmod.ShardedTopK(a.k, dev)
            idx_v, val_v = sh.local_views()
            stream = torch.cuda.current_stream().cuda_stream

            def step(i):
                eng.enqueue(dxs[i % a.queries].data_ptr(), idx_v.data_ptr(), val_v.data_ptr(), stream)
                return sh.step()

            for i in range(a.warmup):
                step(i)
            sync_all()
            t0 = time.perf_counter()
            for i in range(a.steps):
                step(i)
            sync_all()
            elapsed = time.perf_counter() - t0
            exchange = "torch.distributed all_gather_into_tensor of 2*K int32 per rank + on-device merge, every step"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        multi_dist = None
        if a.multi_q and os.environ.get("TKSPMV_DIST", "native") == "native" and os.environ.get("TKSPMV_BENCH_DIST_MULTI") == "1":
            # Extension, beside the headline and never part of `value`: the same step with the local passes serving 8
            # queries each (engines created with multi_q). Opt-in (TKSPMV_BENCH_DIST_MULTI=1): it has only been run with one
            # rank (TKSPMV_BENCH_FORCE_DIST=1, 145-165 k queries/s), and nothing untested may stand between the scaling
            # run and its JSON line. Any rank failing to set it up makes every rank skip it.
            try:
                q = max(a.multi_q)
                meng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=a.k, device=local_rank, first_row=rank * a.rows,
                                stream_replicas=a.replicas, multi_q=q)
                mnat = dmod.NativeShardedSpMV(meng, dev)
                ok = torch.ones(1, device=dev)
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] multi-query distributed leg unavailable ({e})", file=sys.stderr)
                meng = mnat = None
                ok = torch.zeros(1, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() != 0:
                mnat.run_many(dxs.data_ptr(), a.queries, a.warmup)
                mnat.synchronize()
                sync_all()
                t1 = time.perf_counter()
                mnat.run_many(dxs.data_ptr(), a.queries, a.steps)
                mnat.synchronize()
                sync_all()
                tm = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                multi_dist = {"queries_per_pass": q, "value": a.steps * world / float(tm.item()), "unit": "queries/s",
                              "global_queries_per_sec": a.steps / float(tm.item()),
                              "note": "the same sharded step, local passes serving several queries each (tkspmv_enqueue_multi "
                                      "inside tkspmv_dist_*); extension, not part of `value`"}
            if mnat is not None:
                mnat.close()
            if meng is not None:
                meng.close()
        prof = eng.profile(dxs.data_ptr(), a.queries, 200)
        kernel_ns = eng.time_queries(dxs.data_ptr(), a.queries, min(max(a.steps, 50), 500))  # as at N = 1
        units = a.steps * world
        extra = {"global_queries_per_sec": a.steps / elapsed,
                 "kernels_us": {"stream": prof["stream_kernel_ns"] / 1e3, "select": prof["select_kernel_ns"] / 1e3,
                                "query_back_to_back": kernel_ns / 1e3},
                 "exchange": exchange}
        if multi_dist:
            extra["multi_query"] = multi_dist

    if rank == 0:
        line = {
            "metric": "queries_per_sec", "value": units / elapsed, "unit": "queries/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.rows}x{a.cols} gamma nnz/row={a.nnz} (nnz={info['nnz']}) K={a.k} fp32, "
                                   f"queries back to back on one stream, cache-defeated ({a.replicas} rotating stream copies)"
                                   + (f", {world} row shards of {a.rows} rows, RCCL all-gather of K pairs" if world > 1 else ""),
                       "rows": a.rows, "cols": a.cols, "nnz": int(info["nnz"]), "k": a.k,
                       "parallelism": f"row-shard x{world}" if world > 1 else "single GPU",
                       "launch": {"grid": info["grid"], "block": info["block"] + 64,
                                  "wave_partitions": info["n_wave_partitions"], "packet_entries": info["packet_entries"]}},
            "roofline": {"bound": "hbm", "achieved": alg_bytes / kernel_ns, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes / kernel_ns / HBM_PEAK_GBS, "traffic": _traffic_from_profiles(),
                         "kernel": "tkspmv::batch_kernel<4,1024,0> (up to 32 queries per launch; figures are per query)",
                         "algorithmic_bytes": int(alg_bytes),
                         "kernel_us": kernel_ns / 1e3,
                         "method": "one hipEvent pair on the engine stream around a batch of back-to-back launches, "
                                   "duration = batch time / queries. A launch of the batch kernel streams the matrix once "
                                   "per query for up to 32 queries (one continuous prefetch pipeline per wave) and selects "
                                   "each query's top-k in its selector workgroup; rocprofv3's average launch duration / 32 "
                                   "agrees (profiles/README.md)"},
        }
        line.update(extra)
        if world == 1 and a.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(mod, m, xs, a.k, a.cpu_seconds)
        print(json.dumps(line))
    eng.close()
    if multi:
        import torch.distributed as dist
        dist.destroy_process_group()


def _traffic_from_profiles():
    """HBM bytes per launch of the stream kernel from the rocprofv3 --pmc pass committed under profiles/
    (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction); None when no such pass has been recorded."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("stream_kernel_hbm_bytes_per_launch")
    except Exception:
        return None


if __name__ == "__main__":
    main()
