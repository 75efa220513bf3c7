"""Pins the CPU oracle (oracle/oracle.c) to the reference: against the committed golden vectors that were produced
by the reference's own code (tests/golden/make_golden.py -> oracle/_ref) and, where oracle/_ref is present (build
container), against the reference live on fresh random inputs."""
import glob
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
import re
CASE_FILES = sorted(p for p in glob.glob(os.path.join(GOLD, "gold_*.npz")) if re.search(r"_\d+x\d+\.npz$", p))


def _cases(path):
    z = np.load(path)
    return z, json.loads(bytes(z["cases"]).decode())


@pytest.mark.parametrize("path", CASE_FILES, ids=[os.path.basename(p) for p in CASE_FILES])
def test_gold_topk_matches_reference_vectors(oracle, path):
    z, cases = _cases(path)
    assert cases
    for c in cases:
        t, k = c["tag"], c["k"]
        x = z[f"x_{t}"]
        ui, uv = oracle.gold_topk(z["row"], z["col"], z["val"], x, k, sort=False)
        assert np.array_equal(ui, z[f"uidx_{t}"]) and np.array_equal(uv.view(np.uint32), z[f"uval_{t}"].view(np.uint32))
        si, sv = oracle.gold_topk(z["row"], z["col"], z["val"], x, k, sort=True)
        assert np.array_equal(si, z[f"idx_{t}"]) and np.array_equal(sv.view(np.uint32), z[f"val_{t}"].view(np.uint32))


@pytest.mark.parametrize("path", CASE_FILES, ids=[os.path.basename(p) for p in CASE_FILES])
def test_sequential_scores_match_reference_spmv_gold(oracle, path):
    z, cases = _cases(path)
    rows = int(z["rows"])
    for c in cases:
        t = c["tag"]
        y, present = oracle.scores_f32_seq(z["row"], z["col"], z["val"], z[f"x_{t}"], rows)
        assert np.array_equal(y.view(np.uint32), z[f"y_{t}"].view(np.uint32))
        # the selection with the sort_tuples order reproduces the gold list whenever the gold list is full of
        # distinct positive scores (no zero-score / filler corner case)
        k = c["k"]
        gi, gv = z[f"idx_{t}"], z[f"val_{t}"]
        if np.all(gv > 0) and len(np.unique(gv)) == k:
            si, sv = oracle.select_topk(y, present, k)
            assert np.array_equal(si, gi) and np.array_equal(sv, gv)


def test_sample_vector_matches_reference(oracle):
    z = np.load(os.path.join(GOLD, "gold_sample_vector.npz"))
    for size in (16, 1024):
        for sd in (1, 7, 123):
            assert np.array_equal(oracle.sample_vector(size, False, True, sd), z[f"norm_{size}_{sd}"])
            assert np.array_equal(oracle.sample_vector(size, True, False, sd), z[f"sum_{size}_{sd}"])


def test_sort_tuples_matches_reference(oracle):
    z = np.load(os.path.join(GOLD, "gold_eval.npz"))
    si, sv = oracle.sort_tuples(z["idx"], z["val"])
    assert np.array_equal(si, z["sorted_idx"]) and np.array_equal(sv, z["sorted_val"])


@pytest.mark.parametrize("C", [4, 8])
@pytest.mark.parametrize("path", CASE_FILES, ids=[os.path.basename(p) for p in CASE_FILES])
def test_packed_order_model_agrees_with_gold(pkg, oracle, path, C):
    """The order-matched model of the kernel arithmetic: same top-K as the gold (set; order too unless near-ties),
    scores within the north-star tolerance 1e-4 relative."""
    z, cases = _cases(path)
    m = pkg.CooMatrix(int(z["rows"]), int(z["cols"]), z["row"], z["col"], z["val"])
    packed = pkg.Packed(m, k=100, nnz_per_lane=C, n_wave_partitions=37)
    for c in cases:
        t, k = c["tag"], c["k"]
        x = z[f"x_{t}"]
        yp, present = oracle.packed_scores(packed.raw(), x, m.rows, C)
        ys, pres_s = oracle.scores_f32_seq(m.row, m.col, m.val, x, m.rows)
        assert np.array_equal(present, pres_s)
        assert np.allclose(yp, ys, rtol=1e-4, atol=0)
        gi, gv = z[f"idx_{t}"], z[f"val_{t}"]
        if np.all(gv > 0) and len(np.unique(gv)) == k:
            pi, pv = oracle.select_topk(yp, present, k)
            assert set(pi.tolist()) == set(gi.tolist())
            assert np.allclose(pv, gv, rtol=1e-4, atol=0)


def test_select_topk_pads_like_gold(oracle):
    """k larger than the number of rows: the gold's zero-initialised list leaves (0, 0.0) fillers (gold :203-206)."""
    z, cases = _cases(os.path.join(GOLD, "gold_tiny_33x64.npz"))
    c = [c for c in cases if c["k"] == 100][0]
    t = c["tag"]
    gi, gv = z[f"idx_{t}"], z[f"val_{t}"]
    rows = int(z["rows"])
    y, present = oracle.scores_f32_seq(z["row"], z["col"], z["val"], z[f"x_{t}"], rows)
    si, sv = oracle.select_topk(y, present, 100)
    assert np.array_equal(sv, gv) and np.array_equal(si, gi)
    assert np.all(sv[rows:] == 0) and np.all(si[rows:] == 0)


def test_cpu_baseline_port_matches_scipy(pkg, oracle):
    """sparse_dot_topn is absent here (parity unpinned at that boundary); for an N x 1 right-hand side its result is
    A @ x restricted to scores > lower_bound, which scipy reproduces on the same csr_matrix((val,(x,y)))."""
    from scipy.sparse import csr_matrix
    m = pkg.generate_matrix(5000, 512, 20, "gamma", 31)
    x = pkg.create_sample_vector(512, True, False, True, 3).astype(np.float64)
    ptr, idx, v = oracle.coo_to_csr_f64(m.row, m.col, m.val, m.rows)
    A = csr_matrix((m.val.astype(np.float64), (m.row.astype(np.int64), m.col.astype(np.int64))), shape=(m.rows, m.cols))
    A.sum_duplicates()
    assert np.array_equal(A.indptr.astype(np.uint64), ptr) and np.array_equal(A.indices.astype(np.uint32), idx)
    assert np.allclose(A.data, v, rtol=1e-15)
    ref = A @ x
    for threads in (1, 3, 8):
        s, kept = oracle.cpu_topn(ptr, idx, v, m.rows, x, 0.0, threads)
        assert np.allclose(s, np.where(ref > 0, ref, 0.0), rtol=1e-12, atol=0)
        assert np.array_equal(kept.astype(bool), ref > 0)
    gi, gv = oracle.cpu_global_topk(s, kept, 100)
    order = np.lexsort((-np.arange(m.rows), -ref))[:100]
    assert np.array_equal(gi, order.astype(np.uint32))
    # and the fp64 path finds the same top-100 as the fp32 gold on this input
    fi, _ = oracle.gold_topk(m.row, m.col, m.val, x.astype(np.float32), 100)
    assert set(fi.tolist()) == set(gi.tolist())
    s32 = oracle.cpu_spmv_f32(ptr, idx, v.astype(np.float32), m.rows, x.astype(np.float32), 4)
    assert np.allclose(s32, ref, rtol=1e-5)


def test_oracle_against_live_reference(pkg, oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference): the golden-vector tests above cover this")
    for seed in range(6):
        rows, cols = [(300, 64), (1500, 256), (4000, 1024)][seed % 3]
        m = pkg.generate_matrix(rows, cols, 10 + 5 * (seed % 4), ["gamma", "uniform"][seed % 2], 100 + seed)
        for k in (1, 8, 100):
            x = oracle.ref_sample_vector(cols, True, False, True, 40 + seed)
            assert np.array_equal(x, oracle.sample_vector(cols, False, True, 40 + seed))
            ri, rv = oracle.ref_gold_topk(m.row, m.col, m.val, x, k)
            oi, ov = oracle.gold_topk(m.row, m.col, m.val, x, k)
            assert np.array_equal(ri, oi) and np.array_equal(rv.view(np.uint32), ov.view(np.uint32))


def test_segmented_order_is_the_gold_order_for_short_rows(pkg, oracle):
    """oracle_scores_f32_segmented (the multi-query kernel's summation order) restates the gold's sequential fp32 sum
    for every row of at most `seg` entries; a longer row is cut into ceil(len / seg) nearly equal segments (length rounded
    up to a multiple of 4) whose sums are added left to right."""
    m = pkg.generate_matrix(20000, 1024, 20, "gamma", 6)
    x = pkg.create_sample_vector(1024, True, False, True, 3)
    lens = np.bincount(m.row, minlength=m.rows)
    assert lens.max() > 64  # the matrix has rows of both kinds
    y_seq, p_seq = oracle.scores_f32_seq(m.row, m.col, m.val, x, m.rows)
    y_seg, p_seg = oracle.scores_f32_segmented(m.row, m.col, m.val, x, m.rows, 64)
    assert np.array_equal(p_seq, p_seg)
    short = lens <= 64
    assert np.array_equal(y_seq[short].view(np.uint32), y_seg[short].view(np.uint32))
    assert np.allclose(y_seq, y_seg, rtol=1e-6, atol=0)
    y_all, _ = oracle.scores_f32_segmented(m.row, m.col, m.val, x, m.rows, int(lens.max()))
    assert np.array_equal(y_all.view(np.uint32), y_seq.view(np.uint32))
    # a long row by hand: 8 entries with seg = 3 -> ceil(8 / 3) = 3 segments of ceil(8 / 3) = 3, rounded up to 4: 4 + 4
    row = np.zeros(8, np.uint32)
    col = np.arange(8, dtype=np.uint32)
    val = np.array([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8], np.float32)
    xx = np.ones(8, np.float32)
    y3, _ = oracle.scores_f32_segmented(row, col, val, xx, 1, 3)
    f = np.float32
    want = f(f(f(f(f(0.1) + f(0.2)) + f(0.3)) + f(0.4)) + f(f(f(f(0.5) + f(0.6)) + f(0.7)) + f(0.8)))
    assert y3[0].view(np.uint32) == want.view(np.uint32)
