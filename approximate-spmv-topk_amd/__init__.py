"""MI355X-native Top-K SpMV engine: host-side mirror of the reference's operator interface for the hot path.

The compute path is hand-written HIP (csrc/engine.hip) behind the C ABI in include/tkspmv.h; this package only
marshals pointers. There is no CPU fallback.
"""
from . import _lib
from ._lib import TkspmvError, set_option, get_option, options, F32, Q1_7, Q1_7_WIDE, F16, FIXED, Q1_7_F32, MAX_COLS, MAX_K
from .host import CooMatrix, Options, Packed, create_sample_vector, generate_degrees, generate_matrix, generate_matrix_rows, read_mtx, sell_pack_device_check, sell_roundtrip, write_mtx
from .engine import SpMV, topk_spmv

__all__ = ["SpMV", "topk_spmv", "set_option", "get_option", "options", "CooMatrix", "Options", "Packed", "create_sample_vector", "generate_matrix", "generate_matrix_rows", "generate_degrees",
           "read_mtx", "write_mtx", "sell_roundtrip", "sell_pack_device_check", "TkspmvError", "F32", "Q1_7", "Q1_7_WIDE", "F16", "FIXED", "Q1_7_F32", "MAX_COLS", "MAX_K"]


def device_count():
    return _lib.lib().tkspmv_device_count()
