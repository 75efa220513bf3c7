"""The batched exchange of the multi-GPU step with more than one rank's worth of data (row (e) of SURVEY section 8; the host-side merge
being lifted is host_spmv_bscsr.cpp:399-448, global ids :415).

  * tkspmv_merge_topk_batch on a [world][n_q][2][k] buffer filled by REAL shard engines -- the north-star 1M x 1024 matrix
    cut into 2, 4 and 8 nnz-balanced shards, batches of 1, 7 and 32 queries through each shard's batch kernel -- every
    merged list against the gold over the whole matrix;
  * a 2-process rehearsal of the pipelined step (tkspmv_dist_*): two ranks on ONE GPU, the all-gather replaced by a
    host-staged copy through torch.distributed/gloo (RCCL refuses two ranks on one device), while the exchange batches,
    partial batches, buffer rotation, events and merge launches are the real ones.
"""
import os
import subprocess
import sys
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same_as_gold(oracle, m, x, k, idx, val):
    gi, gv = oracle.gold_topk(m.row, m.col, m.val, x, k)
    if set(idx.tolist()) != set(gi.tolist()):  # only k-th-boundary near-ties may differ (2e-6 relative)
        y = oracle.scores_f64(m.row, m.col, m.val, x, m.rows)
        kth = np.sort(y)[-k]
        for r in set(idx.tolist()) ^ set(gi.tolist()):
            assert abs(y[r] - kth) <= 2e-6 * abs(kth), (r, y[r], kth)
    assert np.allclose(np.sort(val)[::-1], np.sort(gv)[::-1], rtol=1e-4, atol=0)
    assert np.all(val[:-1] >= val[1:])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_batched_merge_of_real_shards_against_the_gold(pkg, oracle, world):
    import torch
    dmod = import_module("approximate_spmv_topk_amd.distributed")
    k, rows, n_x = 100, 1000000, 32
    m = pkg.generate_matrix(rows, 1024, 20, "gamma", 2)  # BASELINE configs[1]: what bench.py --gpus N cuts
    bounds = dmod.shard_bounds_by_nnz(m.row, m.rows, world)
    xs = np.stack([pkg.create_sample_vector(1024, True, False, True, 7000 + i) for i in range(n_x)])
    dxs = torch.from_numpy(xs).cuda()
    # every shard's lists for all 32 queries, through the batch kernel (the local step of an exchange batch)
    local = torch.zeros(world, n_x, 2, k, dtype=torch.int32, device="cuda")
    for r, (r0, r1) in enumerate(bounds):
        lr, lc, lv = dmod.shard_coo(m.row, m.col, m.val, r0, r1)
        eng = pkg.SpMV(lr, lc, lv, r1 - r0, 1024, k=k, device=0, first_row=r0)
        out_i = torch.zeros(n_x, k, dtype=torch.int32, device="cuda")
        out_v = torch.zeros(n_x, k, dtype=torch.float32, device="cuda")
        eng.enqueue_batch(dxs.data_ptr(), n_x, out_i.data_ptr(), out_v.data_ptr())
        eng.synchronize()
        local[r, :, 0, :] = out_i
        local[r, :, 1, :] = out_v.view(torch.int32)
        eng.close()
    for n_q in (1, 7, 32):
        gathered = local[:, :n_q].contiguous()  # [world][n_q][2][k]: the layout an all-gather of n_q * 2k words per rank leaves
        mi, mv = dmod.merge_topk_batch_device(gathered.reshape(-1), world, n_q, k)
        torch.cuda.synchronize()
        mi, mv = mi.cpu().numpy().astype(np.uint32), mv.cpu().numpy()
        for q in range(n_q):
            # the single-query merge entry agrees list by list ...
            si, sv = dmod.merge_topk_device(local[:, q].contiguous().reshape(-1), world, k)
            assert np.array_equal(si.cpu().numpy().astype(np.uint32), mi[q]) and np.array_equal(sv.cpu().numpy(), mv[q])
            # ... and every merged list is the gold's over the WHOLE matrix
            if q in (0, n_q // 2, n_q - 1) or n_q <= 7:
                _same_as_gold(oracle, m, xs[q], k, mi[q], mv[q])


def test_two_ranks_on_one_gpu_through_the_pipelined_step(tmp_path):
    """dist_flush with two ranks' worth of data: 2 worker processes, each with the engine of its shard on cuda:0; 75 queries
    in exchange batches of 32 (so a partial batch and the buffer rotation are exercised), every merged list of the last two
    batches checked against the gold over the whole matrix on both ranks."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_host_worker.py"), str(tmp_path)], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert "REHEARSAL_OK" in out, out[-3000:]
