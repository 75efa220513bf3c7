"""ctypes binding of the C ABI declared in include/tkspmv.h (libtkspmv.so, built in-tree by `make`).

There is no Python/CPU fallback: if the shared library is missing the import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TKSPMV_LIB") or os.path.join(_HERE, "libtkspmv.so")  # TKSPMV_LIB: tuning builds

OK, ERR_INVALID, ERR_NOT_SORTED, ERR_DEVICE, ERR_NOMEM, ERR_IO, ERR_UNSUPPORTED, ERR_STATE = range(8)
F32, Q1_7, Q1_7_WIDE, F16, FIXED, Q1_7_F32 = 0, 1, 2, 3, 4, 5
IMPL_STREAM, IMPL_ROW_PER_LANE, IMPL_SCORES_SELECT = 0, 1, 2
MAX_COLS = 16384
MAX_K = 1024


class TkspmvError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"tkspmv error {status}: {message}")
        self.status = status
        self.message = message


class Desc(C.Structure):
    _fields_ = [
        ("rows", C.c_uint32), ("cols", C.c_uint32), ("nnz", C.c_uint64),
        ("row", C.POINTER(C.c_uint32)), ("col", C.POINTER(C.c_uint32)), ("val", C.POINTER(C.c_float)),
        ("k", C.c_int32), ("partitions", C.c_int32), ("k_per_partition", C.c_int32), ("precision", C.c_int32),
        ("device", C.c_int32), ("first_row", C.c_uint32), ("min_score", C.c_float),
        ("waves_per_cu", C.c_int32), ("threads_per_wg", C.c_int32), ("nnz_per_lane", C.c_int32),
        ("stream_replicas", C.c_int32), ("fixed_width", C.c_int32), ("multi_q", C.c_int32), ("impl", C.c_int32), ("reserved", C.c_int32 * 1),
    ]


class Info(C.Structure):
    _fields_ = [
        ("rows", C.c_uint32), ("cols", C.c_uint32), ("nnz", C.c_uint64), ("packed_entries", C.c_uint64),
        ("packed_bytes", C.c_uint64), ("algorithmic_bytes", C.c_uint64), ("n_packets", C.c_uint32),
        ("packet_entries", C.c_uint32), ("n_wave_partitions", C.c_uint32), ("packets_per_partition", C.c_uint32),
        ("grid", C.c_uint32), ("block", C.c_uint32), ("n_groups", C.c_uint32), ("lds_bytes", C.c_uint32),
        ("k", C.c_int32), ("partitions", C.c_int32), ("k_per_partition", C.c_int32), ("precision", C.c_int32),
        ("device", C.c_int32), ("num_cus", C.c_uint32), ("fixed_width", C.c_uint32), ("multi_q", C.c_uint32), ("multi_pack_us", C.c_uint32), ("multi_bytes", C.c_uint64),
        ("pack_us", C.c_uint32), ("pack_on_device", C.c_uint32), ("claim_sets", C.c_uint32), ("batch_mode", C.c_uint32), ("state_bytes", C.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if not n.startswith("reserved")}


class Timing(C.Structure):
    _fields_ = [
        ("stream_kernel_ns", C.c_double), ("select_kernel_ns", C.c_double), ("query_ns", C.c_double),
        ("candidates_avg", C.c_double), ("scores_kernel_ns", C.c_double), ("slow_paths_avg", C.c_double),
        ("appended_avg", C.c_double), ("event_bracket_ns", C.c_double), ("n_queries", C.c_uint32),
        ("reserved", C.c_uint32 * 1),
    ]


class Coo(C.Structure):
    _fields_ = [
        ("rows", C.c_uint32), ("cols", C.c_uint32), ("nnz", C.c_uint64),
        ("row", C.POINTER(C.c_uint32)), ("col", C.POINTER(C.c_uint32)), ("val", C.POINTER(C.c_float)),
        ("num_rows_coo", C.c_uint32), ("index_base", C.c_int32), ("symmetric", C.c_int32),
    ]


class OptionsC(C.Structure):
    _fields_ = [
        ("matrix_path", C.c_char * 1024),
        ("use_sample_matrix", C.c_int32), ("reset", C.c_int32), ("num_tests", C.c_int32), ("debug", C.c_int32),
        ("ignore_matrix_values", C.c_int32), ("top_k_value", C.c_int32),
        ("xclbin_path", C.c_char * 1024),
        ("gpu_impl", C.c_int32), ("use_half_precision_gpu", C.c_int32), ("block_size_1d", C.c_int32),
        ("block_size_2d", C.c_int32), ("num_blocks", C.c_int32),
    ]


# Every symbol include/tkspmv.h declares; tests check the library exports all of them.
EXPORTED_SYMBOLS = [
    "tkspmv_create", "tkspmv_destroy", "tkspmv_get_info", "tkspmv_set_query", "tkspmv_set_query_device",
    "tkspmv_run", "tkspmv_enqueue", "tkspmv_enqueue_many", "tkspmv_enqueue_batch", "tkspmv_synchronize", "tkspmv_read", "tkspmv_result_device", "tkspmv_scores", "tkspmv_debug_trace", "tkspmv_debug_counters",
    "tkspmv_time_queries", "tkspmv_time_host_loop", "tkspmv_time_query_batches", "tkspmv_time_stream_read", "tkspmv_enqueue_multi", "tkspmv_time_multi", "tkspmv_profile", "tkspmv_last_error", "tkspmv_device_count", "tkspmv_mtx_read", "tkspmv_mtx_free",
    "tkspmv_mtx_write", "tkspmv_sample_vector", "tkspmv_generate", "tkspmv_generate_rows", "tkspmv_generate_degrees", "tkspmv_options_parse", "tkspmv_pack", "tkspmv_pack_device",
    "tkspmv_sell_roundtrip", "tkspmv_sell_pack_device_check", "tkspmv_packed_info", "tkspmv_packed_decode", "tkspmv_packed_raw", "tkspmv_packed_free", "tkspmv_wave_partitions", "tkspmv_packed_save", "tkspmv_packed_load",
    "tkspmv_create_packed",
    "tkspmv_dist_unique_id", "tkspmv_dist_create", "tkspmv_dist_set_batch", "tkspmv_dist_enqueue", "tkspmv_dist_run_many",
    "tkspmv_dist_synchronize", "tkspmv_dist_time_exchange", "tkspmv_dist_read", "tkspmv_dist_destroy", "tkspmv_dist_last_error",
    "tkspmv_merge_topk", "tkspmv_merge_topk_batch", "tkspmv_dist_set_host_exchange", "tkspmv_dist_read_batch",
    "tkspmv_set_option", "tkspmv_get_option", "tkspmv_option_count", "tkspmv_option_info",
]

_lib = None


# int (*)(const void *send, void *recv, uint64_t bytes_per_rank, void *user): the host all-gather of the rehearsal exchange
HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


def lib():
    """Loads libtkspmv.so (once). Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). "
            "The Top-K SpMV engine has no Python/CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u32p, f32p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.tkspmv_last_error.restype = C.c_char_p
    L.tkspmv_create.argtypes = [C.POINTER(vp), C.POINTER(Desc)]
    L.tkspmv_destroy.argtypes = [vp]
    L.tkspmv_destroy.restype = None
    L.tkspmv_get_info.argtypes = [vp, C.POINTER(Info)]
    L.tkspmv_set_query.argtypes = [vp, f32p, C.POINTER(C.c_double)]
    L.tkspmv_set_query_device.argtypes = [vp, vp]
    L.tkspmv_run.argtypes = [vp, C.POINTER(C.c_double)]
    L.tkspmv_enqueue.argtypes = [vp, vp, vp, vp, vp]
    L.tkspmv_enqueue_many.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
    L.tkspmv_enqueue_batch.argtypes = [vp, vp, C.c_int32, vp, vp, vp]
    L.tkspmv_synchronize.argtypes = [vp]
    L.tkspmv_read.argtypes = [vp, u32p, f32p, C.POINTER(C.c_int32)]
    L.tkspmv_result_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.tkspmv_scores.argtypes = [vp, f32p]
    L.tkspmv_debug_trace.argtypes = [vp, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64)]
    L.tkspmv_debug_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int32]
    L.tkspmv_time_queries.argtypes = [vp, vp, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    L.tkspmv_time_host_loop.argtypes = [vp, f32p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.tkspmv_time_query_batches.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    L.tkspmv_time_stream_read.argtypes = [vp, C.c_int32, C.POINTER(C.c_double)]
    L.tkspmv_enqueue_multi.argtypes = [vp, vp, C.c_int32, vp, vp, vp]
    L.tkspmv_time_multi.argtypes = [vp, vp, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    L.tkspmv_profile.argtypes = [vp, vp, C.c_int32, C.c_int32, C.POINTER(Timing)]
    L.tkspmv_mtx_read.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Coo)]
    L.tkspmv_mtx_free.argtypes = [C.POINTER(Coo)]
    L.tkspmv_mtx_free.restype = None
    L.tkspmv_mtx_write.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64, u32p, u32p, f32p, C.c_int32,
                                   C.c_int32]
    L.tkspmv_sample_vector.argtypes = [f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.tkspmv_generate.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint64, C.POINTER(Coo)]
    L.tkspmv_generate_rows.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint64, C.POINTER(Coo)]
    L.tkspmv_generate_degrees.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint64, u32p]
    L.tkspmv_options_parse.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(OptionsC)]
    L.tkspmv_pack.argtypes = [C.POINTER(Desc), C.c_uint32, C.POINTER(vp)]
    L.tkspmv_pack_device.argtypes = [C.POINTER(Desc), C.c_uint32, C.POINTER(vp), C.POINTER(C.c_double)]
    L.tkspmv_packed_info.argtypes = [vp, C.POINTER(Info)]
    L.tkspmv_packed_decode.argtypes = [vp, u32p, u32p, f32p, C.POINTER(C.c_uint64)]
    L.tkspmv_packed_raw.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(u32p), C.POINTER(u32p),
                                    C.POINTER(u32p), C.POINTER(C.c_uint32)]
    L.tkspmv_wave_partitions.argtypes = [C.POINTER(Desc), C.POINTER(C.c_uint32)]
    L.tkspmv_packed_save.argtypes = [vp, C.c_char_p]
    L.tkspmv_packed_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.tkspmv_create_packed.argtypes = [C.POINTER(vp), vp, C.POINTER(Desc)]
    L.tkspmv_packed_free.argtypes = [vp]
    L.tkspmv_packed_free.restype = None
    L.tkspmv_dist_last_error.restype = C.c_char_p
    L.tkspmv_dist_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.tkspmv_dist_create.argtypes = [C.POINTER(vp), vp, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]
    L.tkspmv_dist_set_batch.argtypes = [vp, C.c_int32]
    L.tkspmv_dist_enqueue.argtypes = [vp, vp]
    L.tkspmv_dist_run_many.argtypes = [vp, vp, C.c_int32, C.c_int32]
    L.tkspmv_dist_synchronize.argtypes = [vp]
    L.tkspmv_dist_time_exchange.argtypes = [vp, C.c_int32, C.POINTER(C.c_double)]
    L.tkspmv_dist_read.argtypes = [vp, u32p, f32p, C.POINTER(C.c_int32)]
    L.tkspmv_dist_destroy.argtypes = [vp]
    L.tkspmv_dist_destroy.restype = None
    L.tkspmv_merge_topk.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp]
    L.tkspmv_merge_topk_batch.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]
    L.tkspmv_dist_set_host_exchange.argtypes = [vp, HOST_ALLGATHER_FN, vp]
    L.tkspmv_dist_read_batch.argtypes = [vp, u32p, f32p, C.POINTER(C.c_int32)]
    L.tkspmv_set_option.argtypes = [C.c_char_p, C.c_char_p]
    L.tkspmv_get_option.argtypes = [C.c_char_p]
    L.tkspmv_get_option.restype = C.c_char_p
    L.tkspmv_option_info.argtypes = [C.c_int32] + [C.POINTER(C.c_char_p)] * 4
    _lib = L
    return L


def check(status):
    if status != OK:
        raise TkspmvError(status, lib().tkspmv_last_error().decode("utf-8", "replace"))


def set_option(name, value):
    """tkspmv_set_option: an engine option (include/tkspmv.h; `options()` lists them) for the engines created after this call.
    value None: back to unset. Raises TkspmvError for a name that is not in the library's table."""
    check(lib().tkspmv_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name):
    v = lib().tkspmv_get_option(name.encode())
    return None if v is None else v.decode()


def options():
    """The library's table of options: [{name, kind, values, doc, value}]."""
    L = lib()
    out = []
    for i in range(L.tkspmv_option_count()):
        f = [C.c_char_p() for _ in range(4)]
        check(L.tkspmv_option_info(i, *[C.byref(x) for x in f]))
        name, kind, values, doc = (x.value.decode() for x in f)
        out.append({"name": name, "kind": kind, "values": values, "doc": doc, "value": get_option(name)})
    return out


def check_dist(status):
    if status != OK:
        raise TkspmvError(status, lib().tkspmv_dist_last_error().decode("utf-8", "replace"))
