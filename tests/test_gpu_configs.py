"""BASELINE.json configs at their own sizes, through the very paths bench.py measures.

  * configs[1] (1M x 1024, gamma 20, K=100, fp32): the batch kernel with 4 rotating stream copies -- what `value` and
    `roofline` of the bench line are measured on -- against the CPU gold, query by query, and bit for bit against the
    order-matched oracle. Boundary near-ties (the only permitted difference from the gold's index set) are COUNTED and
    printed, not just tolerated.
  * configs[3] (10M x 1024, gamma 20, K=100, 8 row shards): on ONE GPU -- the 10M-row matrix is cut with
    shard_bounds_by_nnz(..., 8), the 8 shards run one after another through engines with desc.first_row, the 8 x K pairs
    go through tkspmv_merge_topk, and the result is compared with the gold over the whole matrix. The reference merge
    this lifts one level up: src/fpga/src/host_spmv_bscsr.cpp:399-448 (`:415` global id = local + first_row).
"""
import os
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-4  # north_star tolerance for fp32 scores
TIE = 2e-6   # relative distance from the k-th score (fp64) within which a row counts as a boundary near-tie


def _compare_with_gold(oracle, m, x, k, idx, val, y64=None):
    """Returns the number of boundary-tie swaps (rows by which the set differs from the gold's); asserts that nothing
    else differs and that the scores agree to RTOL."""
    gi, gv = oracle.gold_topk(m.row, m.col, m.val, x, k)
    swaps = 0
    if set(idx.tolist()) != set(gi.tolist()):
        if y64 is None:
            y64, _ = oracle.scores_f64(m.row, m.col, m.val, x, m.rows)
        kth = np.sort(y64)[-k]
        diff = set(idx.tolist()) ^ set(gi.tolist())
        for r in diff:
            assert abs(y64[r] - kth) <= TIE * kth, f"row {r} differs from the gold and is not a k-th boundary near-tie"
        swaps = len(diff) // 2
    assert np.allclose(np.sort(val)[::-1], np.sort(gv)[::-1], rtol=RTOL, atol=0)
    assert np.all(val[:-1] >= val[1:]) and len(set(idx.tolist())) == k
    return swaps


def test_config1_the_measured_path_against_the_gold_at_full_size(pkg, oracle):
    import torch
    n_q, k = 64, 100
    m = pkg.generate_matrix(1000000, 1024, 20, "gamma", 2)  # bench.py's matrix at N = 1
    xs = np.stack([pkg.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(n_q)])  # bench.py's vectors
    dxs = torch.from_numpy(xs).cuda()
    out_i = torch.zeros(n_q, k, dtype=torch.int32, device="cuda")
    out_v = torch.zeros(n_q, k, dtype=torch.float32, device="cuda")
    eng = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=k, device=0, stream_replicas=4)
    info = eng.info()
    torch.cuda.synchronize()
    eng.enqueue_batch(dxs.data_ptr(), n_q, out_i.data_ptr(), out_v.data_ptr())  # two launches of the batch kernel
    eng.synchronize()
    gi_all = out_i.cpu().numpy().astype(np.uint32)
    gv_all = out_v.cpu().numpy()
    C = info["packet_entries"] // 64
    packed = pkg.Packed(m, k=k, nnz_per_lane=C, n_wave_partitions=(info["batch_mode"] >> 16) or info["n_wave_partitions"])
    assert packed.info()["n_wave_partitions"] == info["n_wave_partitions"]
    raw = packed.raw()
    swaps = 0
    for q in range(n_q):
        idx, val = gi_all[q], gv_all[q]
        swaps += _compare_with_gold(oracle, m, xs[q], k, idx, val)
        yp, present = oracle.packed_scores(raw, xs[q], m.rows, C)
        ei, ev = oracle.select_topk(yp, present, k)
        assert np.array_equal(idx, ei), f"query {q}: index list differs from the order-matched oracle"
        assert np.array_equal(val.view(np.uint32), ev.view(np.uint32)), f"query {q}: scores are not bit-identical"
    print(f"\n[configs[1], batch kernel, 4 stream copies] {n_q} queries x top-{k}: {swaps} boundary-tie swap(s) against the gold")
    # the same queries through tkspmv_enqueue_many (engine-owned buffers: the last query wins), bench.py's timed call
    eng.enqueue_many(dxs.data_ptr(), n_q, n_q)
    eng.synchronize()
    val, idx = eng.read_result()
    assert np.array_equal(idx, gi_all[-1]) and np.array_equal(val.view(np.uint32), gv_all[-1].view(np.uint32))
    eng.close()


def test_config3_ten_million_rows_in_eight_shards_on_one_gpu(pkg, oracle):
    import torch
    dmod = import_module("approximate_spmv_topk_amd.distributed")
    world, k, rows = 8, 100, 10000000
    if os.environ.get("TKSPMV_TEST_SMALL_CFG3"):  # development aid: a tenth of the size
        rows = 1000000
    m = pkg.generate_matrix(rows, 1024, 20, "gamma", 4)  # SURVEY 8(d) cfg 4: seed 4
    bounds = dmod.shard_bounds_by_nnz(m.row, m.rows, world)
    deg = pkg.generate_degrees(0, rows, 20, "gamma", 4)
    assert bounds == dmod.shard_bounds_from_degrees(deg, world)  # what bench.py --gpus N cuts by
    nnz_per = [int(np.searchsorted(m.row, b) - np.searchsorted(m.row, a)) for a, b in bounds]
    assert sum(nnz_per) == m.nnz and max(nnz_per) - min(nnz_per) <= 400
    xs = np.stack([pkg.create_sample_vector(1024, True, False, True, 4000 + i) for i in range(3)])
    dxs = torch.from_numpy(xs).cuda()
    n_q = xs.shape[0]
    gathered = torch.zeros(n_q, world, 2, k, dtype=torch.int32, device="cuda")  # per query: [world][2][k]
    y_packed = [np.zeros(m.rows, np.float32) for _ in range(n_q)]
    for r, (r0, r1) in enumerate(bounds):
        lr, lc, lv = dmod.shard_coo(m.row, m.col, m.val, r0, r1)
        # a rank's shard as bench.py builds it: generated in place, never cut out of the whole matrix
        s = pkg.generate_matrix_rows(r0, r1, 1024, 20, "gamma", 4)
        assert np.array_equal(s.row, lr) and np.array_equal(s.col, lc) and np.array_equal(s.val, lv)
        eng = pkg.SpMV(lr, lc, lv, r1 - r0, 1024, k=k, device=0, first_row=r0)
        info = eng.info()
        for q in range(n_q):
            eng.enqueue(dxs[q].data_ptr(), gathered[q, r, 0].data_ptr(), gathered[q, r, 1].data_ptr())
        eng.synchronize()
        # order-matched scores of this shard (the shard's own packing), for the bit-exact comparison below
        C = info["packet_entries"] // 64
        sm = pkg.CooMatrix(r1 - r0, 1024, lr, lc, lv)
        packed = pkg.Packed(sm, k=k, nnz_per_lane=C, n_wave_partitions=(info["batch_mode"] >> 16) or info["n_wave_partitions"])
        assert packed.info()["n_wave_partitions"] == info["n_wave_partitions"]
        yp, present = oracle.packed_scores(packed.raw(), xs[0], r1 - r0, C)
        assert present.all()
        y_packed[0][r0:r1] = yp
        eng.close()
    swaps = 0
    for q in range(n_q):
        mi, mv = dmod.merge_topk_device(gathered[q].reshape(-1), world, k)
        idx = mi.cpu().numpy().astype(np.uint32)
        val = mv.cpu().numpy()
        swaps += _compare_with_gold(oracle, m, xs[q], k, idx, val)  # the gold over the WHOLE 10M-row matrix
        # the torch-level merge of distributed.py (what the gloo tests cover) agrees with the native merge kernel
        g = gathered[q]
        ti, tv = dmod.merge_candidates(g[:, 0, :].reshape(-1).to(torch.int64) & 0xFFFFFFFF,
                                       g[:, 1, :].reshape(-1).view(torch.float32), k)
        assert np.array_equal(ti.cpu().numpy().astype(np.uint32), idx) and np.array_equal(tv.cpu().numpy(), val)
        if q == 0:  # bit for bit: exact selection over the shards' order-matched scores
            ei, ev = oracle.select_topk(y_packed[0], np.ones(m.rows, np.uint8), k)
            assert np.array_equal(idx, ei) and np.array_equal(val.view(np.uint32), ev.view(np.uint32))
    print(f"\n[configs[3], {rows} rows in {world} nnz-balanced shards, merged on the device] {n_q} queries x top-{k}: "
          f"{swaps} boundary-tie swap(s) against the gold over the whole matrix; shard nnz {min(nnz_per)}..{max(nnz_per)}")


def test_experiment_driver_on_a_small_grid(tmp_path):
    """The counterpart of the reference's test_spmv_topk.py:66-111: tools/run_experiments.py runs the drop-in executable over
    a 2 x 2 grid of generated matrices (reference file naming), keeps the CSVs and writes the accuracy table computed with
    plot_errors.py's metric definitions. fp32 through the default engine variant: every list must equal the CPU gold's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out, mats = tmp_path / "results", tmp_path / "matrices"
    cmd = [sys.executable, os.path.join(root, "tools", "run_experiments.py"), "--rows", "10000", "30000", "--cols", "512",
           "--dist", "gamma", "--nnz", "20", "40", "-k", "100", "-t", "5", "--matrix-folder", str(mats), "--out-folder", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    table = json.load(open(out / "accuracy.json"))
    assert len(table) == 4
    assert sorted((t["rows"], t["nnz"]) for t in table) == [(10000, 20), (10000, 40), (30000, 20), (30000, 40)]
    for t in table:
        assert (mats / f"matrix_{t['rows']}_512_{t['nnz']}_gamma.mtx").exists()  # test_spmv_topk.py:108 naming
        assert (out / t["file"]).exists() and t["iterations"] == 3                # first two iterations dropped (:699)
        assert t["prec_100"] == 1.0
        for th in (1, 8, 16, 32, 50, 75, 100):  # (a prefix may differ by one row where two scores agree to fp32 rounding)
            assert t[f"prec_{th}"] >= 1.0 - 1.0 / th - 1e-9 and t[f"ndcg_{th}"] > 0.99999
        assert t["kendall_100"] > 0.999 and 0 < t["hw_exec_time_ms_mean"] < 5.0
    # the other engine variants (-i 1: one row per lane, -i 2: scores + radix select) return the same lists
    for impl in (1, 2):
        o2 = tmp_path / f"results_i{impl}"
        r = subprocess.run(cmd[:-1] + [str(o2), "--impl", str(impl), "--rows", "10000", "--nnz", "20"], capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        t2 = json.load(open(o2 / "accuracy.json"))
        assert len(t2) == 1 and t2[0]["prec_100"] == 1.0 and t2[0]["kendall_100"] > 0.999


def test_reference_grid_largest_matrix_against_the_gold(pkg, oracle):
    """The far corner of the reference's grid (test_spmv_topk.py:12-47): 15M rows x 1024 columns x 40 non-zeros per row, 600M
    non-zeros -- the first matrix past 2^31 bytes of values and 2^29 entries: tkspmv_run and a batch of queries against the CPU gold
    (gold_algorithms.hpp:188-246 restated and pinned; ~6 s per query on one core, hence two queries)."""
    import torch
    k = 100
    m = pkg.generate_matrix(15000000, 1024, 40, "gamma", 9)
    assert m.nnz > 2 ** 29
    xs = np.stack([pkg.create_sample_vector(1024, True, False, True, 7000 + i) for i in range(4)])
    eng = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=k, device=0)
    assert eng.info()["packed_bytes"] > 2 ** 31
    golds = [oracle.gold_topk(m.row, m.col, m.val, xs[q], k) for q in range(2)]
    for q in range(2):
        eng.reset(xs[q])
        ns = eng()
        val, idx = eng.read_result()
        gi, gv = golds[q]
        assert set(idx.tolist()) == set(gi.tolist()), q
        assert np.allclose(np.sort(val)[::-1], np.sort(gv)[::-1], rtol=1e-4, atol=0)
    dxs = torch.from_numpy(xs).cuda()
    out_i = torch.zeros(4, k, dtype=torch.int32, device="cuda")
    out_v = torch.zeros(4, k, dtype=torch.float32, device="cuda")
    eng.enqueue_batch(dxs.data_ptr(), 4, out_i.data_ptr(), out_v.data_ptr())
    eng.synchronize()
    for q in range(2):
        gi, gv = golds[q]
        assert set(out_i[q].cpu().numpy().astype(np.uint32).tolist()) == set(gi.tolist()), q
    print(f"\n[15M x 1024 x 40: {m.nnz} nnz, {eng.info()['packed_bytes'] / 1e9:.2f} GB packed] tkspmv_run {ns / 1e3:.0f} us per query")
    eng.close()
