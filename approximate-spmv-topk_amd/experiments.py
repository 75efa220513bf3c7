"""Experiment driver pieces and accuracy metrics.

Counterpart of the reference's result post-processing: the executables print one CSV line per iteration with the CPU
gold's list (`sw_res_idx/sw_res_val`) and the accelerator's (`hw_res_idx/hw_res_val`)
(host_spmv_topk_csr_gpu.cu:452,466-467; host_spmv_bscsr.cpp:638-691), and
src/resources/python/plotting/plot_errors.py:85-93,182-231 turns them into precision@t, Kendall's tau and NDCG.
Those three definitions are restated here (and pinned against the reference's functions by golden vectors,
tests/golden/make_golden_metrics.py) so that numbers are comparable with the paper's tables.
"""
import csv
import math
import os

import numpy as np

THRESHOLDS = (1, 8, 16, 32, 50, 75, 100)  # the reference evaluates these prefixes (plot_errors.py THRESHOLDS)

GPU_COLUMNS = ["iteration", "error_idx", "error_val", "sw_full_time_ms", "sw_topk_time_ms", "hw_setup_time_ms",
               "hw_spmv_only_time_ms", "hw_exec_time_ms", "readback_time_ms", "k", "sw_res_idx", "sw_res_val",
               "hw_res_idx", "hw_res_val"]


def precision_at(sw_idx, hw_idx, t):
    """|top-t of the gold ∩ top-t of the accelerator| / t (plot_errors.py:86-88)."""
    return len(set(sw_idx[:t]) & set(hw_idx[:t])) / t


def kendall_tau(reference_rank, predicted_rank):
    """Kendall's tau over the union of the two lists (plot_errors.py:182-216): pairs ranked by both lists count +1 when
    the two orders agree and -1 otherwise; the sum is divided by sqrt(pairs ranked by the reference) *
    sqrt(pairs ranked by the prediction)."""
    items = list(set(reference_rank) | set(predicted_rank))
    ref = {it: p for p, it in enumerate(reference_rank)}
    pred = {it: p for p, it in enumerate(predicted_rank)}
    agree = disagree = in_ref = in_pred = 0
    for a in range(len(items)):
        for b in range(a + 1, len(items)):
            i1, i2 = items[a], items[b]
            r = i1 in ref and i2 in ref
            p = i1 in pred and i2 in pred
            in_ref += r
            in_pred += p
            if r and p:
                if (ref[i1] - ref[i2]) * (pred[i1] - pred[i2]) > 0:
                    agree += 1
                else:
                    disagree += 1
    return (agree - disagree) / (math.sqrt(in_ref) * math.sqrt(in_pred))


def ndcg(sw_idx, sw_val, hw_idx, hw_val):
    """NDCG of the accelerator's list with the gold's scores as relevance (plot_errors.py:219-231): a returned row
    that the gold does not contain has relevance 0; discount 1 / log2(position + 2). Returns (ndcg, dcg, idcg)."""
    rel = dict(zip(sw_idx, sw_val))
    dcg = sum(rel.get(idx, 0) / math.log2(i + 2) for i, idx in enumerate(hw_idx))
    idcg = sum(v / math.log2(i + 2) for i, v in enumerate(sw_val))
    return dcg / idcg, dcg, idcg


def bscsr_packet_size(fixed_width):
    """BSCSR_PACKET_SIZE of the reference for a value width (types.hpp:57-79): (512 - 1) // (W + 10 + 4) -- 15 entries per 512-bit
    packet at 20 bits, 13 at 25, 11 at 32."""
    return (512 - 1) // (int(fixed_width) + 10 + 4)


def hls_dataflow_topk(row, scores, rows, k, partitions=32, k_per_list=8, packet_entries=15, limited=4):
    """The candidate set the reference's HLS cores deliver, from EXACT per-row scores (this engine's SpMV-only kernel in the same
    arithmetic: `SpMV(..., precision=FIXED, fixed_width=W).scores()`): a host-side transform of the matrix's row structure.

    The design keeps, per partition of ceil(rows / partitions) rows (host_spmv_bscsr.cpp:133-141), `limited` independent lists
    of k_per_list entries -- one per packet SLOT (spmv_bscsr_top_k_multicore.hpp:331-409): a row that finishes inside a packet
    of packet_entries entries is offered to list 1 + (row ends before it in that packet); a row whose last entry is the
    packet's last is offered to list 0 by the next packet; lists `limited` and beyond do not exist, so the 4th and later rows
    that finish inside one packet are lost, and so is the last row of every partition (its flush is commented out, :396-403).
    The host keeps every list entry with a positive value, one per row id, and sorts (host_spmv_bscsr.cpp:399-448).

    Not modelled here: a packet with MORE than `limited` row segments also drops the products of the segments beyond and
    shifts the row ids the core reports for the rest of its partition (:104-149,246-326); `overfull_packets` counts them (0 on
    the BASELINE matrices at 15 entries per packet) and oracle/hls_model.c restates that part too.

    row: row ids of the row-sorted COO; scores[r]: exact score of row r. Returns (idx, val, info): the merged top-k in
    sort_tuples order and info = {candidates, lost_rows, overfull_packets}."""
    row = np.asarray(row, dtype=np.int64)
    scores = np.asarray(scores, dtype=np.float32)
    per = (int(rows) + partitions - 1) // partitions
    part = row // per
    # position of every entry inside its partition's packet stream
    first_of_part = np.searchsorted(part, np.arange(partitions + 1))
    local = np.arange(row.shape[0], dtype=np.int64) - first_of_part[part]
    pkt = local // packet_entries
    off = local % packet_entries
    is_end = np.ones(row.shape[0], bool)
    is_end[:-1] = row[1:] != row[:-1]
    part_len = (first_of_part[1:] - first_of_part[:-1])[part]
    last_of_packet = (off == packet_entries - 1) | (local == part_len - 1)
    ends = np.flatnonzero(is_end)
    # row ends before this one inside the same packet of the same partition
    gkey = part[ends] * (1 << 40) + pkt[ends]
    first_in_group = np.ones(ends.shape[0], bool)
    first_in_group[1:] = gkey[1:] != gkey[:-1]
    rank_in_packet = np.arange(ends.shape[0]) - np.maximum.accumulate(np.where(first_in_group, np.arange(ends.shape[0]), 0))
    n_packets = (part_len[ends] + packet_entries - 1) // packet_entries
    at_packet_end = last_of_packet[ends]
    slot = np.where(at_packet_end, 0, 1 + rank_in_packet)
    offered = np.where(at_packet_end, pkt[ends] + 1 < n_packets, slot < limited)
    # segments of a packet = the rows that end in it + the unfinished one behind them (unless its last entry ends a row)
    all_key = part * (1 << 40) + pkt
    pk_ids, pk_inv = np.unique(all_key, return_inverse=True)
    ends_per_packet = np.bincount(pk_inv[ends], minlength=pk_ids.shape[0]) if ends.size else np.zeros(pk_ids.shape[0], np.int64)
    trailing = np.ones(pk_ids.shape[0], np.int64)
    trailing[pk_inv[ends[at_packet_end]]] = 0
    overfull = int((ends_per_packet + trailing > limited).sum())
    r_ids = row[ends]
    val = scores[r_ids]
    keep = offered & (val > 0)
    lists = part[ends] * limited + slot
    cand_idx, cand_val = [], []
    if keep.any():
        li, ri, vi = lists[keep], r_ids[keep], val[keep]
        order = np.lexsort((ri, vi, li))  # by list, then value, then row id (ascending)
        li, ri, vi = li[order], ri[order], vi[order]
        # rank from the top inside each list
        idx_in = np.arange(li.shape[0])
        start = np.ones(li.shape[0], bool)
        start[1:] = li[1:] != li[:-1]
        gstart = np.maximum.accumulate(np.where(start, idx_in, 0))
        size = np.bincount(np.cumsum(start) - 1)[np.cumsum(start) - 1]
        from_top = size - 1 - (idx_in - gstart)
        top = from_top < k_per_list
        cand_idx, cand_val = ri[top], vi[top]
    cand_idx = np.asarray(cand_idx, dtype=np.uint32)
    cand_val = np.asarray(cand_val, dtype=np.float32)
    order = np.lexsort((cand_idx, cand_val))[::-1][:k]  # (value desc, row id desc) = sort_tuples
    info = {"candidates": int(cand_idx.shape[0]), "lost_rows": int((~offered).sum()), "overfull_packets": overfull}
    return cand_idx[order], cand_val[order], info


def read_result_csv(path):
    """Rows of a result CSV in the GPU-host schema (the one bin/approximate-spmv-mi355x-topk prints), lists decoded."""
    out = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if set(GPU_COLUMNS) - set(row):
                raise ValueError(f"{path}: not a GPU-host result file (columns {sorted(row)})")
            k = int(row["k"])
            rec = {c: float(row[c]) for c in GPU_COLUMNS[3:9]}
            rec.update(iteration=int(row["iteration"]), error_idx=int(row["error_idx"]), error_val=int(row["error_val"]), k=k,
                       sw_res_idx=[int(x) for x in row["sw_res_idx"].split(";")][:k],
                       sw_res_val=[float(x) for x in row["sw_res_val"].split(";")][:k],
                       hw_res_idx=[int(x) for x in row["hw_res_idx"].split(";")][:k],
                       hw_res_val=[float(x) for x in row["hw_res_val"].split(";")][:k])
            out.append(rec)
    return out


def accuracy(rows, thresholds=THRESHOLDS, skip=2):
    """Mean precision@t / Kendall's tau / NDCG over the iterations of one result file, and the mean and standard
    deviation of hw_exec_time_ms; the first `skip` iterations are dropped like the reference's own summary does
    (host_spmv_bscsr.cpp:699)."""
    rows = rows[skip:] if len(rows) > skip else rows
    res = {"iterations": len(rows)}
    for t in thresholds:
        use = [r for r in rows if r["k"] >= t]
        if not use:
            continue
        res[f"prec_{t}"] = float(np.mean([precision_at(r["sw_res_idx"], r["hw_res_idx"], t) for r in use]))
        res[f"kendall_{t}"] = float(np.mean([kendall_tau(r["sw_res_idx"][:t], r["hw_res_idx"][:t]) for r in use])) if t > 1 else 1.0
        res[f"ndcg_{t}"] = float(np.mean([ndcg(r["sw_res_idx"][:t], r["sw_res_val"][:t], r["hw_res_idx"][:t],
                                                r["hw_res_val"][:t])[0] for r in use]))
    ex = np.array([r["hw_exec_time_ms"] for r in rows])
    res["hw_exec_time_ms_mean"] = float(ex.mean()) if len(ex) else float("nan")
    res["hw_exec_time_ms_std"] = float(ex.std()) if len(ex) else float("nan")
    res["sw_topk_time_ms_mean"] = float(np.mean([r["sw_topk_time_ms"] for r in rows])) if rows else float("nan")
    return res


def matrix_name(rows, cols, nnz, dist):
    """File name convention of the reference's matrices (test_spmv_topk.py:108): matrix_{rows}_{cols}_{nnz}_{dist}.mtx"""
    return f"matrix_{rows}_{cols}_{nnz}_{dist}.mtx"


def result_name(rows, cols, dist, nnz, k, niter, tag="mi355x", bits="f32"):
    """Result file name in the reference's pattern (test_spmv_topk.py:73,80): {t}_{s}_{c}_{d}_{n}_{bits}_..._{K}_{NITER}.csv;
    bits = "f32", "f16" or e.g. "20b" for a fixed-point width, where the reference writes its FPGA build's bit width."""
    return f"{tag}_{rows}_{cols}_{dist}_{nnz}_{bits}_{k}_{niter}.csv"


def default_exe():
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "approximate-spmv-mi355x-topk")
