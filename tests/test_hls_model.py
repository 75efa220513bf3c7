"""The reference's HLS dataflow approximations: oracle/hls_model.c -- a plain-C restatement of
spmv_bscsr_top_k_multicore.hpp:104-149,246-326,331-409 + host_spmv_bscsr.cpp:133-248,399-448 (PARITY UNPINNED: the HLS kernel
needs Xilinx headers) -- against a hand-worked example, and the product's host-side transform (experiments.hls_dataflow_topk:
exact scores + the matrix's row structure) against the restatement."""
from importlib import import_module

import numpy as np
import pytest


def test_hand_worked_partition(pkg, oracle):
    """One partition, packets of 4 entries, 2 slots, lists of 2, fp32 (W = 0), x = ones (every product is 0.5).
    entries' rows:  0 0 1 2 | 2 2 2 3 | 3 4 5 5 | 5 5 5 6        (| = packet boundaries)
    packet 0 = [0 0 1 2] holds 3 segments, more than `limited` = 2: row 0 (segment 0) finishes and is offered to list 1 with local
    id 0; segment 1 (row 1) is the last AGGREGATED segment, so the core takes it for the packet's unfinished last row; segment 2
    (row 2's first entry) lies beyond the limit and its product is dropped. Packet 1 starts with row 2 -- the row of packet 0's
    last entry, so its xf bit is clear -- and the core adds the carried sum (row 1's) to packet 1's first segment: row 1 is never
    offered to any list, row 2 is offered with 0.5 (row 1's) + 1.5 (its three entries of packet 1)."""
    ex = import_module("approximate_spmv_topk_amd.experiments")
    row = np.array([0, 0, 1, 2, 2, 2, 2, 3, 3, 4, 5, 5, 5, 5, 5, 6], np.uint32)
    col = np.zeros(16, np.uint32)
    val = np.full(16, 0.5, np.float32)
    x = np.ones(8, np.float32)
    ci, cv, slot, local = oracle.hls_model_topk(row, col, val, x, 7, P=1, B=4, K=2, limited=2, W=0)
    assert slot[0] == 1 and local[0] == 0
    assert slot[1] == 0xFF  # swallowed by its successor
    assert slot[2] == 1 and local[2] == 1  # (the core's row counter is one behind from here on: row 1 was never counted)
    assert float(cv[ci == 1][0]) == 2.0 if (ci == 1).any() else True  # reported under local id 1 + first_row 0 = "row 1", value 2.0
    assert np.all(np.mod(cv, 0.5) == 0) and cv.max() <= 3.0
    assert slot[6] == 0xFF  # the partition's last row is never flushed
    # the structural transform flags the overfull packet
    y = np.bincount(row, weights=val).astype(np.float32)
    _, _, info = ex.hls_dataflow_topk(row, y, 7, 4, partitions=1, k_per_list=2, packet_entries=4, limited=2)
    assert info["overfull_packets"] >= 1
    # ... and models it when asked to: the same merged list as the restated cores, fp32 arithmetic
    mi, mv, minfo = ex.hls_dataflow_topk(row, None, 7, 64, partitions=1, k_per_list=2, packet_entries=4, limited=2, overfull="model",
                                         col=col, val=val, vec=x, fixed_width=0)
    assert np.array_equal(mi, ci) and np.array_equal(mv.view(np.uint32), cv.view(np.uint32)) and minfo["overfull_packets"] == info["overfull_packets"]
    # with the matrix's own row ids instead of the core's slipping counter: row 2 is reported as row 2, with the same (wrong) sum
    ti, tv, _ = ex.hls_dataflow_topk(row, None, 7, 64, partitions=1, k_per_list=2, packet_entries=4, limited=2, overfull="model",
                                     col=col, val=val, vec=x, fixed_width=0, ids="matrix")
    assert 1 not in ti.tolist() and float(tv[ti == 2][0]) == 2.0


@pytest.mark.parametrize("rows,nnz,dist,seed,W,P,K", [(20000, 20, "gamma", 3, 20, 32, 8), (9000, 20, "uniform", 5, 25, 32, 8),
                                                      (30000, 24, "gamma", 7, 32, 16, 8), (5000, 20, "gamma", 9, 20, 4, 16)])
def test_transform_equals_the_restated_dataflow(pkg, oracle, rows, nnz, dist, seed, W, P, K):
    """Integer arithmetic (sums independent of their order): with the scores of the integer model the host-side transform
    delivers exactly the merged list of the restated dataflow -- the same candidates, the same order, the same bits --
    wherever no packet holds more than LIMITED_FINISHED_ROWS row segments (it says so otherwise)."""
    ex = import_module("approximate_spmv_topk_amd.experiments")
    m = pkg.generate_matrix(rows, 1024, nnz, dist, seed)
    x = pkg.create_sample_vector(1024, True, False, True, seed + 100)
    B = ex.bscsr_packet_size(W)
    assert B == {20: 15, 25: 13, 32: 11}[W]
    ci, cv, slot, local = oracle.hls_model_topk(m.row, m.col, m.val, x, m.rows, P, B, K, 4, W)
    y, present = oracle.fixed_scores(m.row, m.col, m.val, x, m.rows, W)
    ei, ev, info = ex.hls_dataflow_topk(m.row, y, m.rows, 100, partitions=P, k_per_list=K, packet_entries=B, limited=4)
    assert info["overfull_packets"] == 0  # (20+ entries per row, at most 15 per packet)
    assert info["candidates"] == ci.shape[0] <= P * 4 * K
    assert np.array_equal(ci[:100], ei) and np.array_equal(cv[:100].view(np.uint32), ev.view(np.uint32))
    # rows the cores never offer: the last row of every partition (its flush is commented out in the reference)
    per = (m.rows + P - 1) // P
    last_rows = [int(m.row[np.searchsorted(m.row // per, p, side="right") - 1]) for p in range(P)]
    assert all(slot[r] == 0xFF for r in last_rows) and info["lost_rows"] == int((slot[np.unique(m.row)] == 0xFF).sum())
    # local row ids: partition-local positions among the rows that own entries
    r = int(ci[0])
    assert local[r] == np.searchsorted(np.unique(m.row[m.row // per == r // per]), r)


def test_short_rows_overfill_packets_and_the_transform_says_so(pkg, oracle):
    """4 entries per row on average: packets of 15 entries routinely hold more than 4 segments; the restated dataflow loses
    those rows (and shifts ids), the structural transform reports how many packets are affected instead of pretending."""
    ex = import_module("approximate_spmv_topk_amd.experiments")
    m = pkg.generate_matrix(4000, 1024, 4, "gamma", 11)
    x = pkg.create_sample_vector(1024, True, False, True, 3)
    ci, cv, slot, local = oracle.hls_model_topk(m.row, m.col, m.val, x, m.rows, 8, 15, 8, 4, 20)
    y, _ = oracle.fixed_scores(m.row, m.col, m.val, x, m.rows, 20)
    _, _, info = ex.hls_dataflow_topk(m.row, y, m.rows, 100, partitions=8, k_per_list=8, packet_entries=15, limited=4)
    assert info["overfull_packets"] > 100
    assert (slot[np.unique(m.row)] == 0xFF).sum() > 100  # many rows are never offered to any list


@pytest.mark.parametrize("rows,nnz,dist,seed,W,P,K,limited", [
    (4000, 4, "gamma", 11, 20, 8, 8, 4), (4000, 4, "gamma", 11, 0, 8, 8, 4), (4000, 3, "uniform", 12, 25, 4, 8, 2),
    (20000, 20, "gamma", 3, 20, 32, 8, 4), (3000, 2, "uniform", 5, 32, 3, 4, 3), (5000, 6, "gamma", 9, 12, 16, 16, 1),
    (50000, 8, "gamma", 1, 20, 32, 8, 4)])
def test_limited_finished_rows_modelled_equals_the_restated_dataflow(pkg, oracle, rows, nnz, dist, seed, W, P, K, limited):
    """overfull = "model": LIMITED_FINISHED_ROWS as the cores implement it (products of the segments beyond dropped, the last
    aggregated segment carried, the row counter slipping) -- the product's vectorised transform against the plain-C restatement,
    list for list and bit for bit, on matrices where hundreds to thousands of packets are overfull; fp32 (W = 0) included."""
    ex = import_module("approximate_spmv_topk_amd.experiments")
    m = pkg.generate_matrix(rows, 1024, nnz, dist, seed)
    x = pkg.create_sample_vector(1024, True, False, True, seed + 100)
    B = ex.bscsr_packet_size(W) if W else 11
    ci, cv, slot, local = oracle.hls_model_topk(m.row, m.col, m.val, x, m.rows, P, B, K, limited, W)
    ei, ev, info = ex.hls_dataflow_topk(m.row, None, m.rows, 4096, partitions=P, k_per_list=K, packet_entries=B, limited=limited,
                                        overfull="model", col=m.col, val=m.val, vec=x, fixed_width=W)
    assert np.array_equal(ci, ei) and np.array_equal(cv.view(np.uint32), ev.view(np.uint32))
    assert info["candidates"] == ci.shape[0] and info["lost_rows"] == int((slot[np.unique(m.row)] == 0xFF).sum())
    if nnz < 10:
        assert info["overfull_packets"] > 100


@pytest.mark.parametrize("W,K", [(8, 8), (9, 4), (20, 8)])
def test_modelled_dataflow_with_empty_rows_and_tied_scores(pkg, oracle, W, K):
    """Rows without entries (the cores number the rows they SEE, so every reported id behind one is off) and 8/9-bit arithmetic
    (dozens of equal scores at every list's K-th place: which of them survive depends on the positions they landed in -- those
    lists are replayed entry by entry)."""
    ex = import_module("approximate_spmv_topk_amd.experiments")
    m = pkg.generate_matrix(30000, 1024, 5, "gamma", 21)
    keep = (m.row % 7 != 3) & (m.row % 11 != 0)
    r, c, v = m.row[keep], m.col[keep], m.val[keep]
    x = pkg.create_sample_vector(1024, True, False, True, 5)
    B = ex.bscsr_packet_size(W)
    ci, cv, slot, local = oracle.hls_model_topk(r, c, v, x, 30000, 16, B, K, 4, W)
    ei, ev, info = ex.hls_dataflow_topk(r, None, 30000, 4096, partitions=16, k_per_list=K, packet_entries=B, limited=4, overfull="model",
                                        col=c, val=v, vec=x, fixed_width=W)
    assert np.array_equal(ci, ei) and np.array_equal(cv.view(np.uint32), ev.view(np.uint32))


def test_streamed_list_replay_matches_a_hand_trace():
    """K = 3, offers 5 5 5 5 7 5: the list is [5 5 5] after three offers, the fourth replaces position 0 (first minimum), 7 replaces
    position 0 again, the last 5 replaces position 1: ids 5, 6 and 3 survive -- not 'the latest' and not 'the largest ids'."""
    ex = import_module("approximate_spmv_topk_amd.experiments")
    res, idx = ex._streamed_list(np.array([5, 5, 5, 5, 7, 5]), np.array([1, 2, 3, 4, 5, 6]), 3)
    assert res == [7, 5, 5] and idx == [5, 6, 3]
    l, v, i = ex._hls_lists(np.zeros(6, np.int64), np.array([5, 5, 5, 5, 7, 5], np.uint64), np.array([1, 2, 3, 4, 5, 6]), 3)
    assert sorted(i.tolist()) == [3, 5, 6]


def test_model_mode_needs_the_entries():
    ex = import_module("approximate_spmv_topk_amd.experiments")
    with pytest.raises(ValueError):
        ex.hls_dataflow_topk(np.zeros(4, np.uint32), None, 1, 1, overfull="model")
    with pytest.raises(ValueError):
        ex.hls_dataflow_topk(np.zeros(4, np.uint32), np.ones(1, np.float32), 1, 1, overfull="drop")
