"""Row-sharded Top-K SpMV over several GPUs (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The reference is single-device; its only scaling axis is the row partitioning of the matrix (32 partitions -> HBM
channels, host_spmv_bscsr.cpp:133-141) with a host-side merge of the per-partition candidates
(host_spmv_bscsr.cpp:399-448). Here the same idea is applied one level up:

  * rows are split into contiguous shards balanced by nnz; every rank owns one engine over its shard and returns
    GLOBAL row ids (desc.first_row, cf. `local + first_row` at host_spmv_bscsr.cpp:415);
  * the query x (<= 64 KiB) is replicated;
  * ONE exchange step per query: an all-gather of k (row, score) pairs per rank (800 B at k=100) over xGMI;
  * the final merge of world*k candidates keeps the sort_tuples order (score desc, row desc).
    Each shard returns its full local top-k, so the union contains the global top-k (exactness, cf. topk_errors.py).

The merge is written against plain tensors so the same code runs on CPU with gloo in the tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds_by_nnz(row, rows, world):
    """Contiguous row ranges [r0, r1) per rank with ~equal nnz. `row` is the sorted COO row array."""
    row = np.asarray(row)
    nnz = row.shape[0]
    bounds = [0]
    for r in range(1, world):
        target = (nnz * r) // world
        if nnz == 0:
            bounds.append((rows * r) // world)
            continue
        cut_row = int(row[min(target, nnz - 1)])  # the row containing the target nnz starts the next shard
        cut_row = max(cut_row, bounds[-1])
        bounds.append(min(cut_row, rows))
    bounds.append(rows)
    return [(bounds[i], bounds[i + 1]) for i in range(world)]


def shard_bounds_from_degrees(deg, world):
    """The same cut as shard_bounds_by_nnz, from the row lengths alone (deg[r] = entries of row r): the row that holds
    entry number nnz * r / world starts shard r. Lets every rank of a row-sharded job derive all shard bounds without
    any rank holding the matrix (bench.py --gpus N, BASELINE configs[3])."""
    deg = np.asarray(deg, dtype=np.int64)
    rows = int(deg.shape[0])
    ends = np.cumsum(deg)  # ends[r] = entries in rows [0, r]
    nnz = int(ends[-1]) if rows else 0
    bounds = [0]
    for r in range(1, world):
        if nnz == 0:
            bounds.append((rows * r) // world)
            continue
        target = min((nnz * r) // world, nnz - 1)
        cut_row = int(np.searchsorted(ends, target, side="right"))  # row containing entry `target`
        bounds.append(min(max(cut_row, bounds[-1]), rows))
    bounds.append(rows)
    return [(bounds[i], bounds[i + 1]) for i in range(world)]


def generate_shard(total_rows, cols, avg_nnz, distribution, seed, rank, world):
    """Rank `rank`'s shard of the synthetic total_rows-row matrix, cut by nnz into `world` contiguous row ranges: returns
    (CooMatrix with local row ids, (r0, r1), global nnz). Only the row lengths of the whole matrix are computed; the
    shard's entries are generated in place (every row has its own PRNG streams, host_utils.cpp)."""
    from . import host
    deg = host.generate_degrees(0, total_rows, avg_nnz, distribution, seed)
    bounds = shard_bounds_from_degrees(deg, world)
    r0, r1 = bounds[rank]
    return host.generate_matrix_rows(r0, r1, cols, avg_nnz, distribution, seed), (r0, r1), int(deg.sum(dtype=np.int64))


def shard_coo(row, col, val, r0, r1):
    """Entries of rows [r0, r1), with local row ids."""
    row = np.asarray(row)
    lo = int(np.searchsorted(row, r0, side="left"))
    hi = int(np.searchsorted(row, r1, side="left"))
    return (row[lo:hi] - np.uint32(r0)).astype(np.uint32), np.asarray(col)[lo:hi], np.asarray(val)[lo:hi]


def _order_key(val_f32):
    """Monotone map float32 -> int64 in [0, 2^32): larger score <=> larger key (same map as the kernels)."""
    u = val_f32.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    neg = (u & 0x80000000) != 0
    return torch.where(neg, (~u) & 0xFFFFFFFF, u | 0x80000000)


def merge_candidates(idx_i64, val_f32, k):
    """Top-k of the gathered candidates in sort_tuples order (score desc, row desc). Filler entries (0, 0.0)
    produced by shards with fewer than k qualifying rows sort last among equal scores and are kept only if needed.
    idx_i64: int64 [n] global row ids, val_f32: float32 [n]."""
    key = (_order_key(val_f32) << 32) | (idx_i64 & 0xFFFFFFFF)
    # de-duplicate fillers: several shards may contribute (0, 0.0); a real row id appears in one shard only
    key_sorted, order = torch.sort(key, descending=True)
    keep = torch.ones_like(key_sorted, dtype=torch.bool)
    keep[1:] = key_sorted[1:] != key_sorted[:-1]
    order = order[keep][:k]
    out_idx = idx_i64[order]
    out_val = val_f32[order]
    if out_idx.shape[0] < k:  # pad like the gold's zero-initialised list
        pad = k - out_idx.shape[0]
        out_idx = torch.cat([out_idx, torch.zeros(pad, dtype=out_idx.dtype, device=out_idx.device)])
        out_val = torch.cat([out_val, torch.zeros(pad, dtype=out_val.dtype, device=out_val.device)])
    return out_idx, out_val


class ShardedTopK:
    """Exchange + merge step. `local_idx` / `local_val` are this rank's k results (global row ids)."""

    def __init__(self, k, device, group=None):
        self.k = k
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = device
        # one packed buffer per rank: [k] int32 row ids followed by [k] float32 scores viewed as int32
        self.local = torch.zeros(2 * k, dtype=torch.int32, device=device)
        self.gathered = torch.zeros(self.world * 2 * k, dtype=torch.int32, device=device)

    def local_views(self):
        """(idx_view int32[k], val_view float32[k]) into the send buffer; engines write their results here."""
        return self.local[: self.k], self.local[self.k:].view(torch.float32)

    def exchange(self):
        """The one collective of the path: all-gather of k pairs per rank."""
        if self.world == 1:
            self.gathered.copy_(self.local)
        else:
            dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
        return self.gathered

    def merge(self):
        g = self.gathered.view(self.world, 2, self.k)
        idx = (g[:, 0, :].reshape(-1).to(torch.int64)) & 0xFFFFFFFF
        val = g[:, 1, :].reshape(-1).view(torch.float32)
        return merge_candidates(idx, val, self.k)

    def step(self):
        self.exchange()
        return self.merge()


class NativeShardedSpMV:
    """The same step in native code (csrc/dist.hip): local kernel, RCCL all-gather of k pairs per rank and merge
    kernel; queries are exchanged in batches (default 32) on a side stream, overlapping the local step of the next batch.
    A query vector passed to enqueue() must stay valid until the batch is flushed (batch full, synchronize or read). torch.distributed is only used to ship
    rank 0's RCCL unique id. Raises TkspmvError (e.g. ERR_UNSUPPORTED when RCCL cannot be loaded): callers fall back
    to ShardedTopK."""

    def __init__(self, engine, device, group=None, host_exchange=False):
        """host_exchange=True: rehearsal without RCCL (which refuses two ranks on one device): the all-gather of every exchange
        batch goes through host buffers and torch.distributed (any backend, e.g. gloo); batches, buffer rotation, events and the
        merge launch are the real step's (tkspmv_dist_set_host_exchange)."""
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        self.engine = engine
        self.k = engine.k
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        idbuf = (C.c_uint8 * 128)()
        import os
        if host_exchange:
            os.environ["TKSPMV_DIST_NO_NCCL"] = "1"
        if (world > 1 and not host_exchange) or os.environ.get("TKSPMV_DIST_FORCE_NCCL"):
            t = torch.zeros(128, dtype=torch.uint8, device=device)
            if rank == 0:
                _lib.check_dist(_lib.lib().tkspmv_dist_unique_id(idbuf))
                t.copy_(torch.tensor(list(idbuf), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(t, src=0, group=group)
            for i, b in enumerate(t.cpu().tolist()):
                idbuf[i] = b
        self._h = C.c_void_p()
        try:
            _lib.check_dist(_lib.lib().tkspmv_dist_create(C.byref(self._h), engine._h, idbuf, rank, world))
        finally:
            if host_exchange:  # (whatever tkspmv_dist_create did: a later communicator must not inherit the switch)
                os.environ.pop("TKSPMV_DIST_NO_NCCL", None)
        self.world, self.rank = world, rank
        self._cb = None
        if host_exchange:

            def _allgather(send, recv, nbytes, user):  # noqa: ARG001
                try:
                    n = int(nbytes) // 4
                    mine = torch.from_numpy(np.ctypeslib.as_array((C.c_int32 * n).from_address(send)).copy())
                    outs = [torch.empty(n, dtype=torch.int32) for _ in range(world)]
                    if world > 1:
                        dist.all_gather(outs, mine, group=group)
                    else:
                        outs[0] = mine
                    np.ctypeslib.as_array((C.c_int32 * (n * world)).from_address(recv))[:] = torch.cat(outs).numpy()
                    return 0
                except Exception:  # noqa: BLE001
                    return 1

            self._cb = _lib.HOST_ALLGATHER_FN(_allgather)
            _lib.check_dist(_lib.lib().tkspmv_dist_set_host_exchange(self._h, self._cb, None))

    def set_batch(self, batch):
        """Queries per exchange (1..32, default 32): one local sequence, one all-gather, one merge launch per batch."""
        self._lib.check_dist(self._lib.lib().tkspmv_dist_set_batch(self._h, int(batch)))

    def enqueue(self, dev_x_ptr):
        self._lib.check_dist(self._lib.lib().tkspmv_dist_enqueue(self._h, self._C.c_void_p(int(dev_x_ptr))))

    def run_many(self, dev_xs_ptr, n_x, count):
        self._lib.check_dist(self._lib.lib().tkspmv_dist_run_many(self._h, self._C.c_void_p(int(dev_xs_ptr)), n_x, count))

    def synchronize(self):
        self._lib.check_dist(self._lib.lib().tkspmv_dist_synchronize(self._h))

    def time_exchange(self, iters):
        """ns per exchange of one full batch (all-gather + merge launch alone; collective: every rank calls it)."""
        ns = self._C.c_double()
        self._lib.check_dist(self._lib.lib().tkspmv_dist_time_exchange(self._h, int(iters), self._C.byref(ns)))
        return ns.value

    def read(self):
        C = self._C
        idx = np.empty(self.k, dtype=np.uint32)
        val = np.empty(self.k, dtype=np.float32)
        n = C.c_int32()
        self._lib.check_dist(self._lib.lib().tkspmv_dist_read(self._h, idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                             val.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)))
        return val, idx

    def read_batch(self):
        """(val [n_q][k], idx [n_q][k]) of the most recently exchanged batch (flushes the open one first)."""
        C = self._C
        idx = np.empty((32, self.k), dtype=np.uint32)
        val = np.empty((32, self.k), dtype=np.float32)
        n = C.c_int32()
        self._lib.check_dist(self._lib.lib().tkspmv_dist_read_batch(self._h, idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                                   val.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)))
        return val[:n.value], idx[:n.value]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lib().tkspmv_dist_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk_batch_device(gathered_i32, world, n_q, k, stream=0):
    """The native merge kernel as the pipelined step launches it: gathered_i32 is a [world][n_q][2][k] int32 CUDA tensor (what
    the all-gather of an exchange batch leaves behind); returns (idx int32[n_q][k], val float32[n_q][k])."""
    import ctypes as C
    from . import _lib
    out_idx = torch.zeros((n_q, k), dtype=torch.int32, device=gathered_i32.device)
    out_val = torch.zeros((n_q, k), dtype=torch.float32, device=gathered_i32.device)
    _lib.check_dist(_lib.lib().tkspmv_merge_topk_batch(C.c_void_p(gathered_i32.data_ptr()), world, n_q, k,
                                                       C.c_void_p(out_idx.data_ptr()), C.c_void_p(out_val.data_ptr()),
                                                       C.c_void_p(int(stream))))
    return out_idx, out_val


def merge_topk_device(gathered_i32, world, k, stream=0):
    """The native merge kernel on a [world][2][k] int32 CUDA tensor; returns (idx int32[k], val float32[k]) tensors."""
    import ctypes as C
    from . import _lib
    out_idx = torch.zeros(k, dtype=torch.int32, device=gathered_i32.device)
    out_val = torch.zeros(k, dtype=torch.float32, device=gathered_i32.device)
    _lib.check_dist(_lib.lib().tkspmv_merge_topk(C.c_void_p(gathered_i32.data_ptr()), world, k,
                                                 C.c_void_p(out_idx.data_ptr()), C.c_void_p(out_val.data_ptr()),
                                                 C.c_void_p(int(stream))))
    return out_idx, out_val
