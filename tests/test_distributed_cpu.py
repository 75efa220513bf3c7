"""The N > 1 path on CPU: world_size 2 over gloo. Each rank owns a row shard, computes its local top-k with the
oracle (stand-in for the per-rank HIP engine, which needs a GPU), then runs the real exchange + merge code of
approximate-spmv-topk_amd/distributed.py. The merged result must equal the oracle's global top-k."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, k, seed, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _pkg
        import oracle_lib as O
        from importlib import import_module
        mod = _pkg.load()
        dmod = import_module("approximate_spmv_topk_amd.distributed")
        m = mod.generate_matrix(3000, 256, 12, "gamma", seed)
        x = mod.create_sample_vector(256, True, False, True, seed + 1)
        bounds = dmod.shard_bounds_by_nnz(m.row, m.rows, world)
        r0, r1 = bounds[rank]
        lr, lc, lv = dmod.shard_coo(m.row, m.col, m.val, r0, r1)
        y, present = O.scores_f32_seq(lr, lc, lv, x, r1 - r0)
        li, lvv = O.select_topk(y, present, k, 0.0, first_row=r0)  # what the rank's engine returns (global ids)
        sh = dmod.ShardedTopK(k, torch.device("cpu"))
        iv, vv = sh.local_views()
        iv.copy_(torch.from_numpy(li.astype(np.int64)).to(torch.int32))
        vv.copy_(torch.from_numpy(lvv))
        gi, gv = sh.step()
        yg, pg = O.scores_f32_seq(m.row, m.col, m.val, x, m.rows)
        ei, ev = O.select_topk(yg, pg, k)
        ok = np.array_equal(gi.numpy().astype(np.uint32), ei) and np.array_equal(gv.numpy(), ev)
        # shards are contiguous, cover every row once and are nnz-balanced
        cover = sum(b - a for a, b in bounds) == m.rows and all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
        nnz_per = [int(((m.row >= a) & (m.row < b)).sum()) for a, b in bounds]
        balanced = max(nnz_per) - min(nnz_per) <= 400
        q.put((rank, bool(ok), bool(cover), bool(balanced)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,seed", [(8, 1), (100, 2), (1000, 3)])
def test_sharded_topk_gloo_world2(k, seed):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7 * seed) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, k, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), "merged top-k differs from the global oracle"
    assert all(r[2] and r[3] for r in res)


def _worker_generated(rank, world, port, k, seed, q):
    """bench.py --gpus N's way of building the job (BASELINE configs[3]): no rank ever holds the whole matrix; each derives
    every shard's bounds from the row lengths and generates its own rows in place."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _pkg
        import oracle_lib as O
        from importlib import import_module
        mod = _pkg.load()
        dmod = import_module("approximate_spmv_topk_amd.distributed")
        total_rows, cols, nnz = 70000, 512, 20
        shard, (r0, r1), total_nnz = dmod.generate_shard(total_rows, cols, nnz, "gamma", seed, rank, world)
        x = mod.create_sample_vector(cols, True, False, True, seed + 1)
        y, present = O.scores_f32_seq(shard.row, shard.col, shard.val, x, r1 - r0)
        li, lvv = O.select_topk(y, present, k, 0.0, first_row=r0)
        sh = dmod.ShardedTopK(k, torch.device("cpu"))
        iv, vv = sh.local_views()
        iv.copy_(torch.from_numpy(li.astype(np.int64)).to(torch.int32))
        vv.copy_(torch.from_numpy(lvv))
        gi, gv = sh.step()
        # the checker (this test only) does build the whole matrix
        m = mod.generate_matrix(total_rows, cols, nnz, "gamma", seed)
        lo, hi = np.searchsorted(m.row, r0), np.searchsorted(m.row, r1)
        same_rows = (np.array_equal(shard.row + np.uint32(r0), m.row[lo:hi]) and np.array_equal(shard.col, m.col[lo:hi])
                     and np.array_equal(shard.val, m.val[lo:hi]) and total_nnz == m.nnz)
        same_cut = dmod.shard_bounds_by_nnz(m.row, m.rows, world)[rank] == (r0, r1)
        ei, ev = O.gold_topk(m.row, m.col, m.val, x, k)
        ok = np.array_equal(gi.numpy().astype(np.uint32), ei) and np.array_equal(gv.numpy(), ev)
        q.put((rank, bool(ok), bool(same_rows), bool(same_cut)))
    finally:
        dist.destroy_process_group()


def test_generated_shards_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31700 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_generated, args=(r, 2, port, 100, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), "merged top-k of the generated shards differs from the gold over the whole matrix"
    assert all(r[2] for r in res), "a generated shard differs from the corresponding rows of the whole matrix"
    assert all(r[3] for r in res), "shard bounds from the row lengths differ from shard_bounds_by_nnz"


def test_merge_pads_and_orders():
    sys.path.insert(0, ROOT)
    import _pkg
    from importlib import import_module
    _pkg.load()
    dmod = import_module("approximate_spmv_topk_amd.distributed")
    idx = torch.tensor([5, 9, 0, 0, 7, 3], dtype=torch.int64)
    val = torch.tensor([0.5, 0.5, 0.0, 0.0, 0.9, 0.1], dtype=torch.float32)
    i, v = dmod.merge_candidates(idx, val, 4)
    assert i.tolist() == [7, 9, 5, 3] and np.allclose(v.tolist(), [0.9, 0.5, 0.5, 0.1])
    i, v = dmod.merge_candidates(idx, val, 8)  # fillers collapse, then padding
    assert i.tolist() == [7, 9, 5, 3, 0, 0, 0, 0] and v.tolist()[4:] == [0.0] * 4
