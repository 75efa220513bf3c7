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


def _to_fixed(v, width):
    """ap_ufixed<W,1,AP_TRN_ZERO> of a float as an integer in units of 2^-(W-1): negatives and NaN become 0, values of 2.0 and
    above saturate at 2^W - 1, everything else is truncated (the conversion the reference's packer applies, types.hpp:57-79)."""
    v = np.asarray(v, dtype=np.float64)
    d = np.where(v > 0, v, 0.0) * float(1 << (width - 1))
    return np.minimum(np.floor(d), float((1 << width) - 1)).astype(np.uint64)


def _streamed_list(values, ids, k_per_list):
    """One K-list of the reference's cores, entry by entry (spmv_bscsr_top_k_multicore.hpp:331-409): a zero-initialised list, an
    offer replaces the current worst when it is >= it, the worst is the FIRST minimum. Returns the list's (values, ids)."""
    res = [0] * k_per_list
    idx = [0] * k_per_list
    worst, worst_val = 0, 0
    for v, i in zip(values.tolist(), ids.tolist()):
        if v >= worst_val:
            res[worst], idx[worst] = v, i
            worst_val = min(res)
            worst = res.index(worst_val)
    return res, idx


def _hls_lists(list_id, values, ids, k_per_list):
    """The content of every K-list after its offers (given in arrival order; list_id sorted stably by the caller is not required).
    Lists whose K-th value is shared by more offers than there are places left are replayed entry by entry (which of the equal
    offers survive depends on the positions they landed in); all others are the K largest offers. Returns (list_id, value, id)
    of the surviving entries, grouped by list."""
    if values.shape[0] == 0:
        return list_id, values, ids
    arrival = np.arange(values.shape[0])
    order = np.lexsort((arrival, list_id))
    li, vi, ri = list_id[order], values[order], ids[order]
    start = np.ones(li.shape[0], bool)
    start[1:] = li[1:] != li[:-1]
    g = np.cumsum(start) - 1
    gstart = np.flatnonzero(start)
    # rank from the top by value inside each list
    o2 = np.lexsort((vi, g))
    size = np.bincount(g)
    pos = np.arange(li.shape[0]) - gstart[g[o2]]
    from_top = size[g[o2]] - 1 - pos
    keep_sorted = from_top < k_per_list
    keep = np.zeros(li.shape[0], bool)
    keep[o2[keep_sorted]] = True
    # the K-th value of the lists that overflow, and how many offers share it
    full = size > k_per_list
    kth = np.zeros(size.shape[0], dtype=vi.dtype)
    sel = from_top == k_per_list - 1
    kth[g[o2][sel]] = vi[o2][sel]
    at_kth = (vi == kth[g]) & full[g]
    n_at = np.bincount(g, weights=at_kth, minlength=size.shape[0])
    n_above = np.bincount(g, weights=(vi > kth[g]) & full[g], minlength=size.shape[0])
    replay = np.flatnonzero(full & (n_at > k_per_list - n_above))
    out_l, out_v, out_i = [li[keep & ~np.isin(g, replay)]], [vi[keep & ~np.isin(g, replay)]], [ri[keep & ~np.isin(g, replay)]]
    for gg in replay:
        a, b = gstart[gg], gstart[gg] + size[gg]
        res, idx = _streamed_list(vi[a:b], ri[a:b], k_per_list)
        out_l.append(np.full(k_per_list, li[a], dtype=li.dtype))
        out_v.append(np.asarray(res, dtype=vi.dtype))
        out_i.append(np.asarray(idx, dtype=ri.dtype))
    return np.concatenate(out_l), np.concatenate(out_v), np.concatenate(out_i)


def hls_dataflow_topk(row, scores, rows, k, partitions=32, k_per_list=8, packet_entries=15, limited=4, overfull="count",
                      col=None, val=None, vec=None, fixed_width=0, ids="core"):
    """The candidate set the reference's HLS cores deliver: a host-side transform of the matrix's row structure.

    The design keeps, per partition of ceil(rows / partitions) rows (host_spmv_bscsr.cpp:133-141), `limited` independent lists
    of k_per_list entries -- one per packet SLOT (spmv_bscsr_top_k_multicore.hpp:331-409): a row that finishes inside a packet
    of packet_entries entries is offered to list 1 + (row ends before it in that packet); a row whose last entry is the
    packet's last is offered to list 0 by the next packet; lists `limited` and beyond do not exist, so the 4th and later rows
    that finish inside one packet are lost, and so is the last row of every partition (its flush is commented out, :396-403).
    The host keeps every list entry with a positive value, one per row id, and sorts (host_spmv_bscsr.cpp:399-448).

    `limited` is LIMITED_FINISHED_ROWS (types.hpp:75-77). What a packet with MORE than `limited` row segments does is chosen by
    `overfull`:

      "count"  the STRUCTURAL reading: every row keeps its exact score (`scores[r]`: this engine's SpMV-only kernel in the same
               arithmetic, `SpMV(..., precision=FIXED, fixed_width=W).scores()`) and its matrix row id; rows beyond the slot lists
               are lost; `overfull_packets` counts the packets where the cores would do more than that.
      "model"  the cores as written (spmv_bscsr_top_k_multicore.hpp:104-149,246-326): only the first `limited` segments of a packet
               are aggregated -- the products of the segments beyond are DROPPED --, the last aggregated segment is taken for the
               packet's unfinished row (so its sum is carried into the next packet: added to that packet's first segment, or offered
               to list 0 if the next packet starts a new row), and the row counter advances by the aggregated segments only, so every
               row id the core reports for the rest of its partition falls behind. Needs the entries (`col`, `val`), the query
               (`vec`) and `fixed_width` (ap_ufixed<W,1> arithmetic: products truncated to W - 1 fraction bits, sums wrapping at
               2.0; 0 = fp32, the USE_FLOAT build); `scores` is not used. ids = "core": the ids the core reports (first row of
               the partition + its own counter: list for list what oracle/hls_model.c restates); ids = "matrix": the matrix row
               behind each offer instead -- the same dataflow with a row counter that does not slip.

    row: row ids of the row-sorted COO; scores[r]: exact score of row r. Returns (idx, val, info): the merged top-k in
    sort_tuples order and info = {candidates, lost_rows, overfull_packets}."""
    if overfull not in ("count", "model"):
        raise ValueError("overfull: 'count' or 'model'")
    if ids not in ("core", "matrix"):
        raise ValueError("ids: 'core' or 'matrix'")
    if not (1 <= limited <= packet_entries):
        raise ValueError("limited: 1 .. packet_entries")
    if overfull == "model":
        if col is None or val is None or vec is None:
            raise ValueError("overfull='model' needs col, val and vec (the products of dropped segments are what is modelled)")
        return _hls_cores_as_written(row, col, val, vec, rows, k, partitions, k_per_list, packet_entries, limited, int(fixed_width), ids)
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
    overfull_n = int((ends_per_packet + trailing > limited).sum())
    r_ids = row[ends]
    val_r = scores[r_ids]
    keep = offered & (val_r > 0)
    lists = part[ends] * limited + slot
    _, cand_val, cand_idx = _hls_lists(lists[keep], val_r[keep], r_ids[keep], k_per_list)
    cand_idx = np.asarray(cand_idx, dtype=np.uint32)
    cand_val = np.asarray(cand_val, dtype=np.float32)
    pos = cand_val > 0
    cand_idx, cand_val = cand_idx[pos], cand_val[pos]
    order = np.lexsort((cand_idx, cand_val))[::-1][:k]  # (value desc, row id desc) = sort_tuples
    info = {"candidates": int(cand_idx.shape[0]), "lost_rows": int((~offered).sum()), "overfull_packets": overfull_n}
    return cand_idx[order], cand_val[order], info


def _hls_cores_as_written(row, col, val, vec, rows, k, partitions, k_per_list, B, limited, W, ids):
    """overfull = "model" of hls_dataflow_topk: the packer's segments (host_spmv_bscsr.cpp:189-246), the aggregation of the
    first `limited` of them (spmv_bscsr_top_k_multicore.hpp:104-149), loop 3's carry and row numbering (:246-326), loop 4's
    lists (:331-409) and read_result's merge (host_spmv_bscsr.cpp:399-448), vectorised over all packets of all partitions."""
    if W != 0 and not (8 <= W <= 32):
        raise ValueError("fixed_width: 0 (fp32) or 8..32")
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    nnz = row.shape[0]
    per = (int(rows) + partitions - 1) // partitions
    part = row // per
    first_of_part = np.searchsorted(part, np.arange(partitions + 1))
    part_len = first_of_part[1:] - first_of_part[:-1]
    n_pk = (part_len + B - 1) // B
    pk_base = np.concatenate(([0], np.cumsum(n_pk)))
    total_pk = int(pk_base[-1])
    gp = pk_base[part] + (np.arange(nnz, dtype=np.int64) - first_of_part[part]) // B  # packet of every entry, numbered through
    # segments: runs of one row inside one packet
    seg_start = np.ones(nnz, bool)
    seg_start[1:] = (row[1:] != row[:-1]) | (gp[1:] != gp[:-1])
    seg_first = np.flatnonzero(seg_start)
    seg_gp = gp[seg_first]
    seg_row = row[seg_first]
    pk_first_seg = np.searchsorted(seg_gp, np.arange(total_pk))
    seg_s = np.arange(seg_first.shape[0], dtype=np.int64) - pk_first_seg[seg_gp]
    nseg = np.bincount(seg_gp, minlength=total_pk)
    nrip = np.minimum(nseg, limited)  # num_rows_in_packet: only the first `limited` segments are looked at
    pk_part = np.repeat(np.arange(partitions), n_pk)
    pk_first_entry = seg_first[pk_first_seg]
    first_pk_of_part = np.zeros(total_pk, bool)
    first_pk_of_part[pk_base[:-1][n_pk > 0]] = True
    # xf: the packet's first entry starts a row other than the one the previous packet ended in; not looked at for packet 0 (:259)
    starts_new = np.zeros(total_pk, bool)
    nf = ~first_pk_of_part
    starts_new[nf] = row[pk_first_entry[nf]] != row[pk_first_entry[nf] - 1]
    # sums of the segments
    if W:
        mask = np.uint64((1 << W) - 1)
        fv = _to_fixed(val, W)
        fx = _to_fixed(vec, W)[col]
        prod = ((fv * fx) >> np.uint64(W - 1)) & mask
        seg_sum = np.add.reduceat(prod, seg_first) & mask
    else:
        prod = np.asarray(val, dtype=np.float32) * np.asarray(vec, dtype=np.float32)[col]
        seg_len = np.diff(np.concatenate((seg_first, [nnz])))
        seg_sum = np.zeros(seg_first.shape[0], np.float32)
        for t in range(int(seg_len.max()) if seg_len.size else 0):  # entry by entry, like the core's adder
            m = seg_len > t
            seg_sum[m] = seg_sum[m] + prod[seg_first[m] + t]

    def add(a, b):
        return ((a + b) & mask) if W else (a + b).astype(np.float32)

    # the carry: the sum of the last AGGREGATED segment, continued through packets that hold one segment of the same row
    last_kept = pk_first_seg + nrip - 1
    carry = seg_sum[last_kept].copy()
    carry_row = seg_row[last_kept]
    cont = (nrip == 1) & ~starts_new & nf
    chain = np.zeros(total_pk, np.int64)  # position of a packet inside a run of continuing packets
    idx_pk = np.arange(total_pk)
    run_start = np.maximum.accumulate(np.where(~cont, idx_pk, 0))
    chain[cont] = (idx_pk - run_start)[cont]
    for d in range(1, int(chain.max()) + 1 if total_pk else 0):
        m = np.flatnonzero(chain == d)
        carry[m] = add(carry[m], carry[m - 1])
    carry_in = np.zeros_like(carry)
    carry_in[nf] = carry[np.flatnonzero(nf) - 1]
    carry_in_row = np.full(total_pk, -1, np.int64)
    carry_in_row[nf] = carry_row[np.flatnonzero(nf) - 1]
    # loop 3's row numbering: the counter advances by the rows the core SAW finishing
    finished_rows = nrip + starts_new - 1
    before = np.cumsum(finished_rows) - finished_rows
    before -= np.repeat(before[pk_base[:-1][n_pk > 0]], n_pk[n_pk > 0])  # restart in every partition
    start_row = before + starts_new
    first_row = np.zeros(partitions, np.int64)
    first_row[n_pk > 0] = row[first_of_part[:-1][n_pk > 0]]
    # offers: list 0 <- the carried row when the packet starts a new one; list j <- segment j - 1 for j < num_rows_in_packet
    o0 = np.flatnonzero(starts_new)
    sj = np.flatnonzero(seg_s < nrip[seg_gp] - 1)
    sj_val = seg_sum[sj].copy()
    joined = (seg_s[sj] == 0) & ~starts_new[seg_gp[sj]]
    sj_val[joined] = add(sj_val[joined], carry_in[seg_gp[sj][joined]])
    o_pk = np.concatenate((o0, seg_gp[sj]))
    o_slot = np.concatenate((np.zeros(o0.shape[0], np.int64), seg_s[sj] + 1))
    o_val = np.concatenate((carry_in[o0], sj_val))
    o_core = np.concatenate((start_row[o0] - 1, start_row[seg_gp[sj]] + seg_s[sj])) + first_row[pk_part[o_pk]]
    o_row = np.concatenate((carry_in_row[o0], seg_row[sj]))
    arrival = np.lexsort((o_slot, o_pk))
    o_pk, o_slot, o_val, o_core, o_row = o_pk[arrival], o_slot[arrival], o_val[arrival], o_core[arrival], o_row[arrival]
    live = o_val > 0  # (an offer of 0 can only replace an entry of 0, and the host skips those)
    o_id = o_core if ids == "core" else o_row
    lists = pk_part[o_pk] * limited + o_slot
    l_id, l_val, l_idx = _hls_lists(lists[live], o_val[live], o_id[live], k_per_list)
    pos = l_val > 0
    l_id, l_val, l_idx = l_id[pos], l_val[pos], l_idx[pos]
    # read_result: lists in order, one entry per id (the first one wins)
    order = np.lexsort((l_id,))
    l_id, l_val, l_idx = l_id[order], l_val[order], l_idx[order]
    _, first = np.unique(l_idx, return_index=True)
    first.sort()
    if W:
        out_val = (l_val[first].astype(np.float64) * math.ldexp(1.0, -(W - 1))).astype(np.float32)
    else:
        out_val = l_val[first].astype(np.float32)
    out_idx = l_idx[first].astype(np.uint32)
    order = np.lexsort((out_idx, out_val))[::-1][:k]
    offered_rows = np.unique(o_row[o_row >= 0])
    info = {"candidates": int(out_idx.shape[0]), "lost_rows": int(np.unique(row).shape[0] - offered_rows.shape[0]),
            "overfull_packets": int((nseg > limited).sum())}
    return out_idx[order], out_val[order], info


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
