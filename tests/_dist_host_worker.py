"""Worker of tests/test_gpu_dist_batch.py::test_two_ranks_on_one_gpu_through_the_pipelined_step: rank RANK of WORLD_SIZE, all on
cuda:0, process group over gloo; the native pipelined step with the all-gather staged through the host."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    mod = _pkg.load()
    from importlib import import_module
    dmod = import_module("approximate_spmv_topk_amd.distributed")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    k, rows, n_x, n_queries = 100, 300000, 40, 75
    shard, (r0, r1), total_nnz = dmod.generate_shard(rows, 1024, 20, "gamma", 2, rank, world)
    xs = np.stack([mod.create_sample_vector(1024, True, False, True, 9000 + i) for i in range(n_x)])
    dxs = torch.from_numpy(xs).to(dev)
    eng = mod.SpMV(shard.row, shard.col, shard.val, shard.rows, shard.cols, k=k, device=0, first_row=r0)
    step = dmod.NativeShardedSpMV(eng, dev, host_exchange=True)
    assert step.world == world
    whole = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
    assert whole.nnz == total_nnz

    def check(q_index, idx, val):
        x = xs[q_index % n_x]
        gi, gv = O.gold_topk(whole.row, whole.col, whole.val, x, k)
        assert set(idx.tolist()) == set(gi.tolist()), f"query {q_index}: index set differs from the gold over the whole matrix"
        assert np.allclose(np.sort(val)[::-1], np.sort(gv)[::-1], rtol=1e-4, atol=0)

    # 75 queries = two full exchange batches + a partial one of 11 (buffer sets 0, 1, 0)
    step.run_many(dxs.data_ptr(), n_x, 64)
    vb, ib = step.read_batch()  # the second batch: queries 32..63
    assert vb.shape[0] == 32
    for j in (0, 13, 31):
        check(32 + j, ib[j], vb[j])
    for i in range(64, n_queries):
        step.enqueue(dxs[i % n_x].data_ptr())
    vb, ib = step.read_batch()  # flushes the open batch of 11
    assert vb.shape[0] == n_queries - 64
    for j in range(vb.shape[0]):
        check(64 + j, ib[j], vb[j])
    v, i = step.read()
    assert np.array_equal(i, ib[-1]) and np.array_equal(v, vb[-1])
    # exchange batches of 1 (every query on its own) and of 5
    for batch, n in ((1, 3), (5, 12)):
        step.set_batch(batch)
        step.run_many(dxs.data_ptr(), n_x, n)
        vb, ib = step.read_batch()
        last_n = n % batch or batch
        assert vb.shape[0] == last_n
        for j in range(last_n):
            check(n - last_n + j, ib[j], vb[j])
    step.close()
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
    print("REHEARSAL_OK", rank)


if __name__ == "__main__":
    main()
