"""Host-side helpers mirroring the reference's common layer, bound to the native implementations in libtkspmv.so.

  Options               <- src/common/utils/options.hpp:37-133
  read_mtx              <- src/common/utils/utils.hpp:474-520 (readMtx) + mmio.hpp
  create_sample_vector  <- src/common/utils/utils.hpp:234-267
  generate_matrix       <- src/resources/python/create_matrices.py:58-128 (distributions; own PRNG)
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib


@dataclass
class CooMatrix:
    rows: int
    cols: int
    row: np.ndarray  # uint32 [nnz], non-decreasing when produced by the generator / a row-major MTX file
    col: np.ndarray  # uint32 [nnz]
    val: np.ndarray  # float32 [nnz]
    num_rows_coo: int = 0
    index_base: int = 0
    symmetric: bool = False

    @property
    def nnz(self):
        return int(self.row.shape[0])


def _coo_from_c(c):
    n = int(c.nnz)
    out = CooMatrix(
        rows=int(c.rows), cols=int(c.cols),
        row=np.ctypeslib.as_array(c.row, shape=(max(n, 1),))[:n].copy(),
        col=np.ctypeslib.as_array(c.col, shape=(max(n, 1),))[:n].copy(),
        val=np.ctypeslib.as_array(c.val, shape=(max(n, 1),))[:n].copy(),
        num_rows_coo=int(c.num_rows_coo), index_base=int(c.index_base), symmetric=bool(c.symmetric))
    _lib.lib().tkspmv_mtx_free(C.byref(c))
    return out


def read_mtx(path, index_base=0, read_values=True, sort=False):
    """readMtx(fname, ..., directed=0, read_values, debug, zero_indexed_file, sort_tuples).

    index_base=0 is the reference's compiled-in behaviour (zero_indexed_file=true at every call site);
    1 reads generator output (create_matrices.py writes 1-based ids); -1 auto-detects.
    Raises TkspmvError(ERR_IO) where the reference prints a message and exit(1)s.
    """
    c = _lib.Coo()
    _lib.check(_lib.lib().tkspmv_mtx_read(str(path).encode(), index_base, int(bool(read_values)), int(bool(sort)),
                                          C.byref(c)))
    return _coo_from_c(c)


def write_mtx(path, m, index_base=1, precision=10):
    row = np.ascontiguousarray(m.row, dtype=np.uint32)
    col = np.ascontiguousarray(m.col, dtype=np.uint32)
    val = np.ascontiguousarray(m.val, dtype=np.float32)
    _lib.check(_lib.lib().tkspmv_mtx_write(
        str(path).encode(), m.rows, m.cols, row.shape[0], row.ctypes.data_as(C.POINTER(C.c_uint32)),
        col.ctypes.data_as(C.POINTER(C.c_uint32)), val.ctypes.data_as(C.POINTER(C.c_float)), index_base, precision))


def create_sample_vector(size, random=False, sum_to_one=True, norm_one=False, seed=0):
    """Same argument order and defaults as the reference; seed == 0 draws from std::random_device."""
    vec = np.empty(size, dtype=np.float32)
    _lib.check(_lib.lib().tkspmv_sample_vector(vec.ctypes.data_as(C.POINTER(C.c_float)), size, int(random),
                                               int(sum_to_one), int(norm_one), int(seed)))
    return vec


def generate_matrix(rows, cols, avg_nnz, distribution="gamma", seed=1):
    dist = {"uniform": 0, "gamma": 1}[distribution]
    c = _lib.Coo()
    _lib.check(_lib.lib().tkspmv_generate(rows, cols, avg_nnz, dist, seed, C.byref(c)))
    return _coo_from_c(c)


def generate_matrix_rows(row_begin, row_end, cols, avg_nnz, distribution="gamma", seed=1):
    """Rows [row_begin, row_end) of generate_matrix(rows >= row_end, ...) with LOCAL row ids: a rank's shard of a
    row-sharded job, built without the whole matrix ever existing in the process."""
    dist = {"uniform": 0, "gamma": 1}[distribution]
    c = _lib.Coo()
    _lib.check(_lib.lib().tkspmv_generate_rows(row_begin, row_end, cols, avg_nnz, dist, seed, C.byref(c)))
    return _coo_from_c(c)


def generate_degrees(row_begin, row_end, avg_nnz, distribution="gamma", seed=1):
    """Row lengths of rows [row_begin, row_end) of the generated matrix (uint32 array)."""
    dist = {"uniform": 0, "gamma": 1}[distribution]
    deg = np.empty(max(row_end - row_begin, 1), dtype=np.uint32)
    _lib.check(_lib.lib().tkspmv_generate_degrees(row_begin, row_end, avg_nnz, dist, seed,
                                                  deg.ctypes.data_as(C.POINTER(C.c_uint32))))
    return deg[:row_end - row_begin]


@dataclass
class Options:
    matrix_path: str
    use_sample_matrix: bool
    reset: bool
    num_tests: int
    debug: int
    ignore_matrix_values: bool
    top_k_value: int
    xclbin_path: str
    gpu_impl: int
    use_half_precision_gpu: bool
    block_size_1d: int
    block_size_2d: int
    num_blocks: int

    @staticmethod
    def parse(argv):
        """argv includes the program name, like main(argc, argv)."""
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        o = _lib.OptionsC()
        _lib.check(_lib.lib().tkspmv_options_parse(len(argv), arr, C.byref(o)))
        return Options(o.matrix_path.decode(), bool(o.use_sample_matrix), bool(o.reset), o.num_tests, o.debug,
                       bool(o.ignore_matrix_values), o.top_k_value, o.xclbin_path.decode(), o.gpu_impl,
                       bool(o.use_half_precision_gpu), o.block_size_1d, o.block_size_2d, o.num_blocks)


def sell_roundtrip(m, n_wave_partitions=4088, precision=_lib.F32):
    """Packs m into the wave-sliced ELL layout of the multi-query kernel and decodes it again (layout tests):
    (row, col, val, info) with the entries grouped by row, rows in stream order; info = dict of the layout's sizes."""
    row = np.ascontiguousarray(m.row, dtype=np.uint32)
    col = np.ascontiguousarray(m.col, dtype=np.uint32)
    val = np.ascontiguousarray(m.val, dtype=np.float32)
    d = _lib.Desc()
    d.rows, d.cols, d.nnz = m.rows, m.cols, row.shape[0]
    d.precision = int(precision)  # Q1_7_F32: byte chunks (Q1.7 rounded to nearest; decoded values are the rounded ones)
    d.row = row.ctypes.data_as(C.POINTER(C.c_uint32))
    d.col = col.ctypes.data_as(C.POINTER(C.c_uint32))
    d.val = val.ctypes.data_as(C.POINTER(C.c_float))
    nn = max(int(d.nnz), 1)
    orow, ocol, oval = np.empty(nn, np.uint32), np.empty(nn, np.uint32), np.empty(nn, np.float32)
    n = C.c_uint64()
    info = (C.c_uint64 * 6)()
    _lib.check(_lib.lib().tkspmv_sell_roundtrip(
        C.byref(d), int(n_wave_partitions), orow.ctypes.data_as(C.POINTER(C.c_uint32)), ocol.ctypes.data_as(C.POINTER(C.c_uint32)),
        oval.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n), info))
    n = int(n.value)
    keys = ("slices", "chunks", "padded_entries", "partitions", "stream_bytes", "most_chunks_per_partition")
    return orow[:n], ocol[:n], oval[:n], dict(zip(keys, (int(v) for v in info)))


def sell_pack_device_check(m, n_wave_partitions=4088, precision=_lib.F32, device=-1):
    """Packs m into the wave-sliced ELL layout on the host and with the device packer (needs a GPU) and compares the two
    byte for byte: dict(identical, stream_bytes, chunks, host_ms, plan_ms, upload_ms, fill_ms)."""
    row = np.ascontiguousarray(m.row, dtype=np.uint32)
    col = np.ascontiguousarray(m.col, dtype=np.uint32)
    val = np.ascontiguousarray(m.val, dtype=np.float32)
    d = _lib.Desc()
    d.rows, d.cols, d.nnz = m.rows, m.cols, row.shape[0]
    d.precision, d.device = int(precision), int(device)
    d.row = row.ctypes.data_as(C.POINTER(C.c_uint32))
    d.col = col.ctypes.data_as(C.POINTER(C.c_uint32))
    d.val = val.ctypes.data_as(C.POINTER(C.c_float))
    info = (C.c_uint64 * 3)()
    ms = (C.c_double * 4)()
    _lib.check(_lib.lib().tkspmv_sell_pack_device_check(C.byref(d), int(n_wave_partitions), info, ms))
    return {"identical": bool(info[0]), "stream_bytes": int(info[1]), "chunks": int(info[2]), "host_ms": ms[0],
            "plan_ms": ms[1], "upload_ms": ms[2], "fill_ms": ms[3]}


class Packed:
    """Host-side packed (wave-BSCSR) matrix, for layout tests: decode(pack(A)) == A."""

    def __init__(self, m, k=100, nnz_per_lane=0, n_wave_partitions=4096, precision=_lib.F32, fixed_width=0, on_device=False,
                 device=-1):
        """on_device=True: packed by the HIP kernels of csrc/device_pack.hip (needs a GPU) instead of the host packer; the
        result is copied back, so decode() / raw() / save() work alike. self.pack_ms = (upload, kernels) then."""
        self._h = C.c_void_p()
        self._row = np.ascontiguousarray(m.row, dtype=np.uint32)
        self._col = np.ascontiguousarray(m.col, dtype=np.uint32)
        self._val = np.ascontiguousarray(m.val, dtype=np.float32)
        d = _lib.Desc()
        d.rows, d.cols, d.nnz = m.rows, m.cols, self._row.shape[0]
        d.row = self._row.ctypes.data_as(C.POINTER(C.c_uint32))
        d.col = self._col.ctypes.data_as(C.POINTER(C.c_uint32))
        d.val = self._val.ctypes.data_as(C.POINTER(C.c_float))
        d.k, d.precision, d.nnz_per_lane, d.fixed_width = k, precision, nnz_per_lane, fixed_width
        d.device = device
        self.pack_ms = None
        if on_device:
            ms = (C.c_double * 2)()
            _lib.check(_lib.lib().tkspmv_pack_device(C.byref(d), n_wave_partitions, C.byref(self._h), ms))
            self.pack_ms = (ms[0], ms[1])
        else:
            _lib.check(_lib.lib().tkspmv_pack(C.byref(d), n_wave_partitions, C.byref(self._h)))
        self.nnz = int(d.nnz)

    @classmethod
    def load(cls, path):
        """A packed matrix read back from a .tkspmv file (raises TkspmvError ERR_IO if it is missing or damaged)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().tkspmv_packed_load(str(path).encode(), C.byref(self._h)))
        self.nnz = self.info()["nnz"]
        return self

    def save(self, path):
        _lib.check(_lib.lib().tkspmv_packed_save(self._h, str(path).encode()))

    @staticmethod
    def wave_partitions(device=-1, waves_per_cu=0, threads_per_wg=0, m=None, precision=_lib.F32, nnz_per_lane=0):
        """Wave partitions an engine on `device` cuts a matrix into (the n_wave_partitions to pack for). Needs a GPU. With the
        matrix `m` given: the hint tkspmv_create itself would use for it (small matrices keep 4 workgroups for selections);
        without: the largest count an engine of this geometry accepts."""
        d = _lib.Desc()
        d.device, d.waves_per_cu, d.threads_per_wg = int(device), int(waves_per_cu), int(threads_per_wg)
        if m is not None:
            d.rows, d.cols, d.nnz, d.precision, d.nnz_per_lane = int(m.rows), int(m.cols), int(m.row.shape[0]), int(precision), int(nnz_per_lane)
        n = C.c_uint32()
        _lib.check(_lib.lib().tkspmv_wave_partitions(C.byref(d), C.byref(n)))
        return int(n.value)

    def info(self):
        i = _lib.Info()
        _lib.check(_lib.lib().tkspmv_packed_info(self._h, C.byref(i)))
        return i.as_dict()

    def decode(self):
        row = np.empty(max(self.nnz, 1), dtype=np.uint32)
        col = np.empty(max(self.nnz, 1), dtype=np.uint32)
        val = np.empty(max(self.nnz, 1), dtype=np.float32)
        n = C.c_uint64()
        _lib.check(_lib.lib().tkspmv_packed_decode(
            self._h, row.ctypes.data_as(C.POINTER(C.c_uint32)), col.ctypes.data_as(C.POINTER(C.c_uint32)),
            val.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)))
        n = int(n.value)
        return row[:n], col[:n], val[:n]

    def raw(self):
        """(packets bytes, packet_bytes, pkt_row, part_first, part_count) as numpy views/copies."""
        pk = C.c_void_p()
        pb = C.c_uint64()
        prow = C.POINTER(C.c_uint32)()
        pf = C.POINTER(C.c_uint32)()
        pc = C.POINTER(C.c_uint32)()
        npart = C.c_uint32()
        _lib.check(_lib.lib().tkspmv_packed_raw(self._h, C.byref(pk), C.byref(pb), C.byref(prow), C.byref(pf),
                                                C.byref(pc), C.byref(npart)))
        info = self.info()
        n_packets, n_parts = info["n_packets"], int(npart.value)
        nbytes = n_packets * int(pb.value)
        packets = np.ctypeslib.as_array(C.cast(pk, C.POINTER(C.c_uint8)), shape=(max(nbytes, 1),))[:nbytes].copy()
        pkt_row = np.ctypeslib.as_array(prow, shape=(max(n_packets, 1),))[:n_packets].copy()
        part_first = np.ctypeslib.as_array(pf, shape=(max(n_parts, 1),))[:n_parts].copy()
        part_count = np.ctypeslib.as_array(pc, shape=(max(n_parts, 1),))[:n_parts].copy()
        return packets, int(pb.value), pkt_row, part_first, part_count

    def close(self):
        if self._h:
            _lib.lib().tkspmv_packed_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
