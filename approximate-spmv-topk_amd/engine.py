"""Python mirror of the reference's engine concept `struct SpMV` (setup in the constructor, then per query
reset(vec) -> operator()() -> read_result()), bound to the HIP engine through the C ABI (include/tkspmv.h).

  SpMV(...)            <- src/gpu/host_spmv_topk_csr_gpu.cu:95-169, src/fpga/src/host_spmv_bscsr.cpp:104-131
  SpMV.__call__(debug) <- operator()(int debug): runs one query, returns the kernel time in ns  (:171 / :323)
  SpMV.read_result()   <- read_result(res, res_idx): (values, indices), value-descending       (:233 / :399)
  SpMV.reset(vec)      <- reset(vec, debug): installs a new query vector, returns ns          (:241 / :450)

The compute path is the HIP library only; constructing an engine without a GPU raises TkspmvError(ERR_DEVICE).
"""
import ctypes as C

import numpy as np

from . import _lib


class SpMV:
    def __init__(self, x, y, val, num_rows, num_cols, num_nnz=None, vec=None, k=20, debug=0, *, device=-1,
                 first_row=0, min_score=0.0, partitions=1, k_per_partition=0, precision=_lib.F32, waves_per_cu=0,
                 threads_per_wg=0, nnz_per_lane=0, stream_replicas=0, fixed_width=0, multi_q=0, impl=0):
        """x, y, val: row-sorted COO (row ids, column ids, values) as the FPGA host passes them
        (host_spmv_bscsr.cpp:585); val=None means all ones (-v). precision=FIXED: the FPGA's fixed-point real_type of
        `fixed_width` bits (8..32; 0 = 32, the reference's FIXED_WIDTH default, types.hpp:20)."""
        self._h = C.c_void_p()
        row = np.ascontiguousarray(x, dtype=np.uint32)
        col = np.ascontiguousarray(y, dtype=np.uint32)
        v = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
        nnz = int(row.shape[0]) if num_nnz is None else int(num_nnz)
        d = _lib.Desc()
        d.rows, d.cols, d.nnz = int(num_rows), int(num_cols), nnz
        d.row = row.ctypes.data_as(C.POINTER(C.c_uint32))
        d.col = col.ctypes.data_as(C.POINTER(C.c_uint32))
        d.val = v.ctypes.data_as(C.POINTER(C.c_float)) if v is not None else None
        d.k, d.partitions, d.k_per_partition, d.precision = int(k), int(partitions), int(k_per_partition), precision
        d.device, d.first_row, d.min_score = int(device), int(first_row), float(min_score)
        d.waves_per_cu, d.threads_per_wg, d.nnz_per_lane = int(waves_per_cu), int(threads_per_wg), int(nnz_per_lane)
        d.stream_replicas = int(stream_replicas)
        d.fixed_width = int(fixed_width)
        d.multi_q = int(multi_q)
        d.impl = int(impl)
        _lib.check(_lib.lib().tkspmv_create(C.byref(self._h), C.byref(d)))
        self.k = int(k)
        self.num_rows, self.num_cols, self.num_nnz = int(num_rows), int(num_cols), nnz
        self.debug = debug
        if vec is not None:
            self.reset(vec)

    @classmethod
    def from_packed(cls, packed, k=20, debug=0, *, vec=None, device=-1, first_row=0, min_score=0.0, precision=None,
                    stream_replicas=0, multi_q=0):
        """Engine straight from a packed matrix (host.Packed, e.g. Packed.load("matrix.tkspmv")): no MatrixMarket
        parsing, no packing. precision: None = the packed value type (F32 / Q1_7 / F16 / FIXED), or Q1_7_WIDE for Q1.7 values."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        info = packed.info()
        d = _lib.Desc()
        d.rows, d.cols, d.nnz = info["rows"], info["cols"], info["nnz"]
        d.k = int(k)
        d.precision = info["precision"] if precision is None else precision
        d.device, d.first_row, d.min_score = int(device), int(first_row), float(min_score)
        d.stream_replicas = int(stream_replicas)
        d.multi_q = int(multi_q)
        _lib.check(_lib.lib().tkspmv_create_packed(C.byref(self._h), packed._h, C.byref(d)))
        self.k = int(k)
        self.num_rows, self.num_cols, self.num_nnz = info["rows"], info["cols"], info["nnz"]
        self.debug = debug
        if vec is not None:
            self.reset(vec)
        return self

    # -- the four verbs ---------------------------------------------------------------------------------
    def reset(self, vec, debug=0):
        v = np.ascontiguousarray(vec, dtype=np.float32)
        if v.shape[0] != self.num_cols:
            raise ValueError(f"query vector has {v.shape[0]} entries, expected {self.num_cols}")
        ns = C.c_double()
        _lib.check(_lib.lib().tkspmv_set_query(self._h, v.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ns)))
        return int(ns.value)

    def __call__(self, debug=0):
        ns = C.c_double()
        _lib.check(_lib.lib().tkspmv_run(self._h, C.byref(ns)))
        return ns.value

    def read_result(self, debug=0):
        idx = np.empty(self.k, dtype=np.uint32)
        val = np.empty(self.k, dtype=np.float32)
        n = C.c_int32()
        _lib.check(_lib.lib().tkspmv_read(self._h, idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                          val.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)))
        return val[:n.value], idx[:n.value]

    # -- extras -----------------------------------------------------------------------------------------
    def reset_device(self, dev_ptr):
        """Query vector already resident in HBM (raw device pointer, e.g. torch tensor.data_ptr())."""
        _lib.check(_lib.lib().tkspmv_set_query_device(self._h, C.c_void_p(int(dev_ptr))))

    def enqueue(self, dev_x=0, dev_idx=0, dev_val=0, stream=0):
        """Asynchronous launch of one query on `stream` (raw hipStream_t handle; 0 = engine stream)."""
        _lib.check(_lib.lib().tkspmv_enqueue(self._h, C.c_void_p(int(dev_x)), C.c_void_p(int(dev_idx)),
                                             C.c_void_p(int(dev_val)), C.c_void_p(int(stream))))

    def enqueue_many(self, dev_xs, n_x, count, stream=0):
        """`count` queries back to back from a device array of n_x query vectors (no host sync)."""
        _lib.check(_lib.lib().tkspmv_enqueue_many(self._h, C.c_void_p(int(dev_xs)), int(n_x), int(count),
                                                  C.c_void_p(int(stream))))

    def time_queries(self, dev_xs, n_x, iters):
        """ns per query of `iters` back-to-back queries (one hipEvent pair around the batch; nothing else launched)."""
        ns = C.c_double()
        _lib.check(_lib.lib().tkspmv_time_queries(self._h, C.c_void_p(int(dev_xs)), int(n_x), int(iters), C.byref(ns)))
        return ns.value

    def time_host_loop(self, host_xs, iters):
        """The reference's reset / operator() / read_result loop run natively `iters` times over the rows of host_xs (float32,
        [n_x, cols]): (loop_us[iters], kernel_us[iters]) -- the host clock around the three calls, and tkspmv_run's own figure."""
        xs = np.ascontiguousarray(host_xs, dtype=np.float32)
        loop = np.zeros(int(iters), dtype=np.float64)
        kern = np.zeros(int(iters), dtype=np.float64)
        _lib.check(_lib.lib().tkspmv_time_host_loop(self._h, xs.ctypes.data_as(C.POINTER(C.c_float)), int(xs.shape[0]), int(iters),
                                                    loop.ctypes.data_as(C.POINTER(C.c_double)), kern.ctypes.data_as(C.POINTER(C.c_double))))
        return loop / 1e3, kern / 1e3

    def time_query_batches(self, dev_xs, n_x, iters, reps):
        """`reps` batches of `iters` back-to-back queries, all enqueued before the first wait: ns per query of every batch (the GPU
        never idles between them: the kernel under sustained load)."""
        out = (C.c_double * int(reps))()
        _lib.check(_lib.lib().tkspmv_time_query_batches(self._h, C.c_void_p(int(dev_xs)), int(n_x), int(iters), int(reps), out))
        return [float(v) for v in out]

    def time_stream_read(self, passes):
        """ns per pass of a kernel that only loads the engine's packet stream (engine geometry, one launch, rotating
        stream copies): the floor this GPU sets for any kernel that streams the matrix (measurement aid)."""
        ns = C.c_double()
        _lib.check(_lib.lib().tkspmv_time_stream_read(self._h, int(passes), C.byref(ns)))
        return ns.value

    def enqueue_batch(self, dev_xs, count, dev_idx=0, dev_val=0, stream=0):
        """A batch of `count` queries (rows of a device array, stride cols floats); query i's k results go to
        dev_idx + i*k / dev_val + i*k (device pointers; 0 => engine buffers, last query wins). No host sync."""
        _lib.check(_lib.lib().tkspmv_enqueue_batch(self._h, C.c_void_p(int(dev_xs)), int(count),
                                                   C.c_void_p(int(dev_idx)) if dev_idx else None,
                                                   C.c_void_p(int(dev_val)) if dev_val else None,
                                                   C.c_void_p(int(stream))))

    def enqueue_multi(self, dev_xs, count, dev_idx=0, dev_val=0, stream=0):
        """enqueue_batch with several queries per pass over the matrix (info()["multi_q"] of them share every chunk that
        is loaded; engine created with multi_q > 0). Same arguments; dev_xs = 0 with count = 1: the vector installed by
        reset(). No host sync."""
        _lib.check(_lib.lib().tkspmv_enqueue_multi(self._h, C.c_void_p(int(dev_xs)) if dev_xs else None, int(count),
                                                   C.c_void_p(int(dev_idx)) if dev_idx else None,
                                                   C.c_void_p(int(dev_val)) if dev_val else None,
                                                   C.c_void_p(int(stream))))

    def time_multi(self, dev_xs, n_x, iters):
        """ns per query of `iters` queries through the multi-query path (one hipEvent pair around the sequence)."""
        ns = C.c_double()
        _lib.check(_lib.lib().tkspmv_time_multi(self._h, C.c_void_p(int(dev_xs)), int(n_x), int(iters), C.byref(ns)))
        return ns.value

    def debug_counters(self):
        """Checked thresholds of back-to-back queries (info()["batch_mode"]): how many selections failed their check so far (and
        sent their query through the repair launch), the suspension state of carried thresholds, batch launches so far."""
        out = (C.c_uint64 * 19)()
        _lib.check(_lib.lib().tkspmv_debug_counters(self._h, out, 19))
        return {"checks_failed": int(out[0]), "suspension_length": int(out[1]), "suspended_for": int(out[2]), "batch_launches": int(out[3]),
                "local_off_for_launches": int(out[4]), "local_off_length": int(out[5]),
                # tkspmv_run through the single-query kernel (local thresholds, checked): launches, queries repeated through the
                # exact launch because their check failed, and the suspension of carried thresholds that follows a failure
                "single_launches": int(out[6]), "single_repairs": int(out[7]), "single_checks_failed": int(out[8]),
                "single_suspended_for": int(out[9]),
                # batch launches that went out without a repair launch behind them (the host looks at their verdicts when it waits),
                # and the repairs that had to follow after all
                "trusted_launches": int(out[10]), "late_repairs": int(out[11]),
                # the pacing of back-to-back queries in force and what tkspmv_create's measurement of it took (0: static default)
                "pace_quantum": int(out[12]) & 0xFF, "pace_levels": (int(out[12]) >> 8) & 0xFF, "pace_base": (int(out[12]) >> 16) & 0xFF,
                "pace_period_ns": int(out[12]) >> 32,
                "pace_tuned_us": int(out[13]) & 0xFFFFFFFF, "pace_tune_launches": int(out[13]) >> 32,
                # option STATS, summed over the time_multi calls so far (the multi-query kernel's threshold exchange): queries, waves that
                # ran into their bounded wait for a threshold and the ticks (10 ns) they spent there, rows offered to / overflowed from the lists
                "multi_stat_queries": int(out[14]), "multi_waits": int(out[15]), "multi_wait_ticks": int(out[16]),
                "multi_rows_offered": int(out[17]), "multi_rows_overflowed": int(out[18])}

    def synchronize(self):
        _lib.check(_lib.lib().tkspmv_synchronize(self._h))

    def scores(self):
        """Full y = A.x of the current query (verification aid; the hot path never materialises it)."""
        y = np.empty(max(self.num_rows, 1), dtype=np.float32)
        _lib.check(_lib.lib().tkspmv_scores(self._h, y.ctypes.data_as(C.POINTER(C.c_float))))
        return y[:self.num_rows]

    def profile(self, dev_xs, n_x, iters):
        t = _lib.Timing()
        _lib.check(_lib.lib().tkspmv_profile(self._h, C.c_void_p(int(dev_xs)), int(n_x), int(iters), C.byref(t)))
        return {n: getattr(t, n) for n, _ in t._fields_ if n != "reserved"}

    def result_device(self):
        a, b = C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().tkspmv_result_device(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def info(self):
        i = _lib.Info()
        _lib.check(_lib.lib().tkspmv_get_info(self._h, C.byref(i)))
        return i.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().tkspmv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def topk_spmv(m, vec, k=100, **kw):
    """One-shot helper: build the engine for CooMatrix m, run one query, return (values, indices)."""
    e = SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=vec, k=k, **kw)
    try:
        e()
        return e.read_result()
    finally:
        e.close()
