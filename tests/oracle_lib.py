"""ctypes bindings of the CPU oracle (oracle/liboracle.so) and, when present, of the reference's own gold compiled
from /root/reference (oracle/_ref/libref_gold.so). Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libref_gold.so")

u32p, f32p, f64p, u8p, u64p = (C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_double),
                              C.POINTER(C.c_uint8), C.POINTER(C.c_uint64))


def _p(a, t):
    return a.ctypes.data_as(t)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_PATH):
            raise RuntimeError(f"{ORACLE_PATH} missing: run `make` (or __graft_entry__.build())")
        _oracle = C.CDLL(ORACLE_PATH)
    return _oracle


def have_ref():
    return os.path.exists(REF_PATH)


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_PATH)
        _ref.ref_mean.restype = C.c_float
        _ref.ref_st_dev.restype = C.c_float
        _ref.ref_coo_num_rows.restype = C.c_uint
    return _ref


# ---- oracle wrappers ------------------------------------------------------------------------------------
def gold_topk(row, col, val, vec, k, sort=True):
    row, col, val, vec = _u32(row), _u32(col), _f32(val), _f32(vec)
    idx = np.zeros(k, dtype=np.uint32)
    out = np.zeros(k, dtype=np.float32)
    fn = oracle().oracle_gold_topk_sorted if sort else oracle().oracle_gold_topk
    fn(_p(row, u32p), _p(col, u32p), _p(val, f32p), C.c_uint64(row.shape[0]), _p(vec, f32p), C.c_int(k),
       _p(idx, u32p), _p(out, f32p))
    return idx, out


def sort_tuples(idx, val):
    idx, val = _u32(idx).copy(), _f32(val).copy()
    oracle().oracle_sort_tuples(C.c_uint64(idx.shape[0]), _p(idx, u32p), _p(val, f32p))
    return idx, val


def scores_f32_seq(row, col, val, vec, rows):
    row, col, val, vec = _u32(row), _u32(col), _f32(val), _f32(vec)
    y = np.zeros(max(rows, 1), dtype=np.float32)
    present = np.zeros(max(rows, 1), dtype=np.uint8)
    oracle().oracle_scores_f32_seq(_p(row, u32p), _p(col, u32p), _p(val, f32p), C.c_uint64(row.shape[0]),
                                   _p(vec, f32p), C.c_uint32(rows), _p(y, f32p), _p(present, u8p))
    return y[:rows], present[:rows]


def scores_f32_segmented(row, col, val, vec, rows, seg=64):
    row, col, val, vec = _u32(row), _u32(col), _f32(val), _f32(vec)
    y = np.zeros(max(rows, 1), dtype=np.float32)
    present = np.zeros(max(rows, 1), dtype=np.uint8)
    oracle().oracle_scores_f32_segmented(_p(row, u32p), _p(col, u32p), _p(val, f32p), C.c_uint64(row.shape[0]), _p(vec, f32p),
                                         C.c_uint32(rows), C.c_uint32(seg), _p(y, f32p), _p(present, u8p))
    return y[:rows], present[:rows]


def scores_f64(row, col, val, vec, rows):
    row, col, val, vec = _u32(row), _u32(col), _f32(val), _f32(vec)
    y = np.zeros(max(rows, 1), dtype=np.float64)
    present = np.zeros(max(rows, 1), dtype=np.uint8)
    oracle().oracle_scores_f64(_p(row, u32p), _p(col, u32p), _p(val, f32p), C.c_uint64(row.shape[0]), _p(vec, f32p),
                               C.c_uint32(rows), _p(y, f64p), _p(present, u8p))
    return y[:rows], present[:rows]


def q17_scores(row, col, val, vec, rows):
    row, col, val, vec = _u32(row), _u32(col), _f32(val), _f32(vec)
    y = np.zeros(max(rows, 1), dtype=np.float32)
    present = np.zeros(max(rows, 1), dtype=np.uint8)
    oracle().oracle_q17_scores(_p(row, u32p), _p(col, u32p), _p(val, f32p), C.c_uint64(row.shape[0]), _p(vec, f32p),
                               C.c_uint32(rows), _p(y, f32p), _p(present, u8p))
    return y[:rows], present[:rows]


def fixed_scores(row, col, val, vec, rows, width):
    row, col, val, vec = _u32(row), _u32(col), _f32(val), _f32(vec)
    y = np.zeros(max(rows, 1), dtype=np.float32)
    present = np.zeros(max(rows, 1), dtype=np.uint8)
    rc = oracle().oracle_fixed_scores(_p(row, u32p), _p(col, u32p), _p(val, f32p), C.c_uint64(row.shape[0]),
                                      _p(vec, f32p), C.c_uint32(rows), C.c_uint32(width), _p(y, f32p), _p(present, u8p))
    assert rc == 0, "fixed width out of range"
    return y[:rows], present[:rows]


def q17_wide_scores(row, col, val, vec, rows):
    row, col, val, vec = _u32(row), _u32(col), _f32(val), _f32(vec)
    y = np.zeros(max(rows, 1), dtype=np.float32)
    present = np.zeros(max(rows, 1), dtype=np.uint8)
    sh = oracle().oracle_q17_wide_scores(_p(row, u32p), _p(col, u32p), _p(val, f32p), C.c_uint64(row.shape[0]),
                                         _p(vec, f32p), C.c_uint32(vec.shape[0]), C.c_uint32(rows), _p(y, f32p),
                                         _p(present, u8p))
    return y[:rows], present[:rows], sh


def select_topk(y, present, k, min_score=0.0, first_row=0):
    y = _f32(y)
    rows = y.shape[0]
    present = np.ascontiguousarray(present, dtype=np.uint8)
    idx = np.zeros(k, dtype=np.uint32)
    out = np.zeros(k, dtype=np.float32)
    oracle().oracle_select_topk(_p(y, f32p), _p(present, u8p), C.c_uint32(rows), C.c_int(k), C.c_float(min_score),
                                C.c_uint32(first_row), _p(idx, u32p), _p(out, f32p))
    return idx, out


def packed_scores(packed, x, rows, C_lane=4):
    """packed = Packed.raw() tuple; returns the bit-exact fp32 scores the fused kernel must produce."""
    packets, packet_bytes, pkt_row, part_first, part_count = packed
    x = _f32(x)
    y = np.zeros(max(rows, 1), dtype=np.float32)
    present = np.zeros(max(rows, 1), dtype=np.uint8)
    packets = np.ascontiguousarray(packets, dtype=np.uint8)
    oracle().oracle_packed_scores(_p(packets, u8p), C.c_uint64(packet_bytes), _p(_u32(pkt_row), u32p),
                                  _p(_u32(part_first), u32p), _p(_u32(part_count), u32p),
                                  C.c_uint32(len(part_first)), C.c_uint32(C_lane), _p(x, f32p), C.c_uint32(rows),
                                  _p(y, f32p), _p(present, u8p))
    return y[:rows], present[:rows]


def round_to_half(values):
    """values rounded to IEEE binary16 (nearest even) and back: the value stream of TKSPMV_F16."""
    v = _f32(values)
    out = np.empty_like(v)
    oracle().oracle_round_values_to_half(_p(v, f32p), _p(out, f32p), C.c_uint64(v.shape[0]))
    return out


def round_to_q17(values):
    """values rounded to Q1.7 bytes (nearest, ties up, saturating) and back: the value stream of TKSPMV_Q1_7_F32."""
    v = _f32(values)
    out = np.empty_like(v)
    oracle().oracle_round_values_to_q17(_p(v, f32p), _p(out, f32p), C.c_uint64(v.shape[0]))
    return out


def sample_vector(size, sum_to_one=False, norm_one=True, seed=1):
    v = np.zeros(size, dtype=np.float32)
    oracle().oracle_sample_vector(_p(v, f32p), C.c_int(size), C.c_int(int(sum_to_one)), C.c_int(int(norm_one)),
                                  C.c_uint32(seed))
    return v


def coo_to_csr_f64(row, col, val, rows):
    row, col, val = _u32(row), _u32(col), _f32(val)
    nnz = row.shape[0]
    ptr = np.zeros(rows + 1, dtype=np.uint64)
    idx = np.zeros(max(nnz, 1), dtype=np.uint32)
    v = np.zeros(max(nnz, 1), dtype=np.float64)
    n_out = C.c_uint64()
    rc = oracle().oracle_coo_to_csr_f64(_p(row, u32p), _p(col, u32p), _p(val, f32p), C.c_uint64(nnz),
                                        C.c_uint32(rows), _p(ptr, u64p), _p(idx, u32p), _p(v, f64p), C.byref(n_out))
    assert rc == 0
    n = int(n_out.value)
    return ptr, idx[:n].copy(), v[:n].copy()


def cpu_topn(ptr, idx, v, rows, x, lower_bound=0.0, n_threads=1):
    x = np.ascontiguousarray(x, dtype=np.float64)
    scores = np.zeros(max(rows, 1), dtype=np.float64)
    kept = np.zeros(max(rows, 1), dtype=np.uint8)
    rc = oracle().oracle_cpu_topn(_p(ptr, u64p), _p(idx, u32p), _p(v, f64p), C.c_uint32(rows), _p(x, f64p),
                                  C.c_double(lower_bound), C.c_int(n_threads), _p(scores, f64p), _p(kept, u8p))
    assert rc == 0
    return scores[:rows], kept[:rows]


def cpu_global_topk(scores, kept, k):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    kept = np.ascontiguousarray(kept, dtype=np.uint8)
    idx = np.zeros(k, dtype=np.uint32)
    val = np.zeros(k, dtype=np.float64)
    oracle().oracle_cpu_global_topk(_p(scores, f64p), _p(kept, u8p), C.c_uint32(scores.shape[0]), C.c_int(k),
                                    _p(idx, u32p), _p(val, f64p))
    return idx, val


def cpu_spmv_f32(ptr, idx, v32, rows, x, n_threads=1):
    x = _f32(x)
    v32 = _f32(v32)
    scores = np.zeros(max(rows, 1), dtype=np.float32)
    rc = oracle().oracle_cpu_spmv_f32(_p(ptr, u64p), _p(idx, u32p), _p(v32, f32p), C.c_uint32(rows), _p(x, f32p),
                                      C.c_int(n_threads), _p(scores, f32p))
    assert rc == 0
    return scores[:rows]


def cpu_bench(ptr, idx, v64, rows, xs, k, n_threads, warm, reps, use_f32=False):
    """(spmv_ms[reps], total_ms[reps]) of the CPU baseline, timed inside the C library (buffers preallocated)."""
    xs64 = np.ascontiguousarray(xs, dtype=np.float64)
    xs32 = np.ascontiguousarray(xs, dtype=np.float32)
    v64 = np.ascontiguousarray(v64, dtype=np.float64)
    v32 = np.ascontiguousarray(v64, dtype=np.float32)
    a = np.zeros(reps, dtype=np.float64)
    b = np.zeros(reps, dtype=np.float64)
    rc = oracle().oracle_cpu_bench(_p(ptr, u64p), _p(idx, u32p), _p(v64, f64p), _p(v32, f32p), C.c_uint32(rows),
                                   _p(xs64, f64p), _p(xs32, f32p), C.c_int(xs64.shape[0]), C.c_uint32(xs64.shape[1]),
                                   C.c_int(k), C.c_int(n_threads), C.c_int(warm), C.c_int(reps), C.c_int(int(use_f32)),
                                   _p(a, f64p), _p(b, f64p))
    assert rc == 0
    return a, b


# ---- reference wrappers (only in the build container, where oracle/_ref was compiled) --------------------
def ref_gold_topk(row, col, val, vec, k, sort=True):
    row, col, val, vec = _u32(row), _u32(col), _f32(val), _f32(vec)
    idx = np.zeros(k, dtype=np.uint32)
    out = np.zeros(k, dtype=np.float32)
    ref().ref_gold_topk(_p(row, u32p), _p(col, u32p), _p(val, f32p), C.c_ulonglong(row.shape[0]), _p(vec, f32p),
                        C.c_int(k), C.c_int(int(sort)), _p(idx, u32p), _p(out, f32p))
    return idx, out


def ref_sample_vector(size, random=True, sum_to_one=False, norm_one=True, seed=1):
    v = np.zeros(size, dtype=np.float32)
    ref().ref_create_sample_vector(_p(v, f32p), C.c_int(size), C.c_int(int(random)), C.c_int(int(sum_to_one)),
                                   C.c_int(int(norm_one)), C.c_int(seed))
    return v


def ref_read_mtx(path, read_values=True, zero_indexed=True):
    rows, cols, nnzh = C.c_uint(), C.c_uint(), C.c_uint()
    n = C.c_ulonglong()
    ref().ref_read_mtx(str(path).encode(), C.c_int(int(read_values)), C.c_int(int(zero_indexed)), C.byref(rows),
                       C.byref(cols), C.byref(nnzh), C.byref(n))
    n = int(n.value)
    row = np.zeros(max(n, 1), dtype=np.uint32)
    col = np.zeros(max(n, 1), dtype=np.uint32)
    val = np.zeros(max(n, 1), dtype=np.float32)
    ref().ref_read_mtx_fetch(_p(row, u32p), _p(col, u32p), _p(val, f32p))
    return rows.value, cols.value, nnzh.value, row[:n], col[:n], val[:n]


def ref_spmv_gold_csr(row, col, val, rows, cols, vec):
    row, col, val, vec = _u32(row), _u32(col), _f32(val), _f32(vec)
    y = np.zeros(max(rows, 1), dtype=np.float32)
    ref().ref_spmv_gold_csr(_p(row, u32p), _p(col, u32p), _p(val, f32p), C.c_ulonglong(row.shape[0]), C.c_uint(rows),
                            C.c_uint(cols), _p(vec, f32p), _p(y, f32p))
    return y[:rows]


def hls_model_topk(row, col, val, vec, rows, P, B, K, limited, W, max_out=4096):
    """The reference's HLS dataflow restated (oracle/hls_model.c): (idx, val) of the merged candidates in sort_tuples order,
    row_slot [rows] (0xFF = never offered to a list), row_local [rows]."""
    nnz = row.shape[0]
    out_idx = np.zeros(max_out, np.uint32)
    out_val = np.zeros(max_out, np.float32)
    slot = np.zeros(rows, np.uint8)
    local = np.zeros(rows, np.uint32)
    f = oracle().hls_model_topk
    f.restype = C.c_int
    n = f(_p(row, u32p), _p(col, u32p), _p(val, f32p), C.c_uint64(nnz), C.c_uint32(rows), _p(vec, f32p), C.c_uint32(P), C.c_uint32(B),
          C.c_uint32(K), C.c_uint32(limited), C.c_uint32(W), _p(out_idx, u32p), _p(out_val, f32p), C.c_uint32(max_out),
          slot.ctypes.data_as(C.POINTER(C.c_uint8)), _p(local, u32p))
    if n < 0:
        raise ValueError("hls_model_topk: bad parameters")
    n = min(n, max_out)
    return out_idx[:n], out_val[:n], slot, local
