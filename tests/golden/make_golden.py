#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by RUNNING THE REFERENCE's own code (oracle/_ref/libref_gold.so,
compiled from the headers under /root/reference by `make ref`). Only works in the build container; the produced
files are data (inputs + the reference's outputs) and are committed so that the pin tests also run where
/root/reference does not exist.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg  # noqa: E402
import oracle_lib as O  # noqa: E402

mod = _pkg.load()
assert O.have_ref(), "run `make ref` first (needs /root/reference)"


def topk_case(name, m, k_list, seeds):
    out = {"rows": m.rows, "cols": m.cols, "row": m.row, "col": m.col, "val": m.val}
    cases = []
    for k in k_list:
        for sd in seeds:
            x = O.ref_sample_vector(m.cols, True, False, True, sd)  # the reference's create_sample_vector
            gi, gv = O.ref_gold_topk(m.row, m.col, m.val, x, k, sort=True)
            ui, uv = O.ref_gold_topk(m.row, m.col, m.val, x, k, sort=False)
            y = O.ref_spmv_gold_csr(m.row, m.col, m.val, m.rows, m.cols, x)
            tag = f"k{k}_s{sd}"
            out[f"x_{tag}"] = x
            out[f"idx_{tag}"] = gi
            out[f"val_{tag}"] = gv
            out[f"uidx_{tag}"] = ui
            out[f"uval_{tag}"] = uv
            out[f"y_{tag}"] = y
            cases.append({"k": k, "seed": sd, "tag": tag})
    out["cases"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, m.rows, m.cols, m.nnz, len(cases), "cases")


def hand_matrix():
    """1-nnz rows, duplicate (row, col) entries, a long row, an all-zero-score row and empty rows in between."""
    rows, cols = 40, 64
    r, c, v = [], [], []
    rng = np.random.RandomState(5)
    for i in range(rows):
        if i in (7, 8, 21):  # empty rows
            continue
        n = 1 if i % 5 == 0 else (60 if i == 13 else int(rng.randint(2, 9)))
        cs = np.sort(rng.randint(0, cols, n))
        if i == 3:
            cs[:] = cs[0]  # every entry on the same column: pure duplicates
        vs = rng.rand(n).astype(np.float32)
        vs /= np.linalg.norm(vs)
        r += [i] * n
        c += cs.tolist()
        v += vs.tolist()
    return mod.CooMatrix(rows, cols, np.array(r, np.uint32), np.array(c, np.uint32), np.array(v, np.float32))


topk_case("gold_gamma_1000x512", mod.generate_matrix(1000, 512, 20, "gamma", 1), [8, 100], [1, 2])
topk_case("gold_uniform_1000x512", mod.generate_matrix(1000, 512, 20, "uniform", 2), [100], [3])
topk_case("gold_tiny_33x64", mod.generate_matrix(33, 64, 5, "uniform", 7), [8, 100], [4])
topk_case("gold_gamma_2000x1024", mod.generate_matrix(2000, 1024, 20, "gamma", 9), [100], [5, 6])
topk_case("gold_hand_40x64", hand_matrix(), [8, 20], [11, 12])

# ---- create_sample_vector ---------------------------------------------------------------------------------------
vec = {}
for size in (16, 1024):
    for sd in (1, 7, 123):
        vec[f"norm_{size}_{sd}"] = O.ref_sample_vector(size, True, False, True, sd)
        vec[f"sum_{size}_{sd}"] = O.ref_sample_vector(size, True, True, False, sd)
vec["ones_sum_8"] = O.ref_sample_vector(8, False, True, False, 0)
np.savez_compressed(os.path.join(HERE, "gold_sample_vector.npz"), **vec)

# ---- readMtx on a small file written by our writer (generator output format: 1-based, 10 significant digits) -------
m = mod.generate_matrix(200, 128, 12, "gamma", 21)
p1 = os.path.join(HERE, "small_1indexed.mtx")
p0 = os.path.join(HERE, "small_0indexed.mtx")
mod.write_mtx(p1, m, index_base=1)
mod.write_mtx(p0, m, index_base=0)
rd = {}
for tag, path, zero in (("one_as_one", p1, False), ("zero_as_zero", p0, True), ("one_as_zero", p1, True)):
    rows, cols, nnzh, r, c, v = O.ref_read_mtx(path, True, zero)
    rd[f"{tag}_hdr"] = np.array([rows, cols, nnzh], np.uint32)
    rd[f"{tag}_row"], rd[f"{tag}_col"], rd[f"{tag}_val"] = r, c, v
    rd[f"{tag}_num_rows_coo"] = np.array([O.ref().ref_coo_num_rows(r.ctypes.data_as(O.u32p), len(r))], np.uint32)
np.savez_compressed(os.path.join(HERE, "gold_read_mtx.npz"), **rd)

# ---- Options ---------------------------------------------------------------------------------------------------------
import ctypes as C  # noqa: E402


class RefOpt(C.Structure):
    _fields_ = mod._lib.OptionsC._fields_


argvs = [
    ["exe"],
    ["exe", "-m", "/data/matrix_10000_1024_20_gamma.mtx", "-k", "100", "-t", "30"],
    ["exe", "-d", "-s", "-v", "-r", "-a", "-b", "64", "-c", "4", "-g", "128", "-i", "1", "-x", "foo.xclbin"],
    ["exe", "--matrix_path", "a.mtx", "--k", "8", "--num_tests", "5", "--no_reset", "--gpu_impl", "2",
     "--half_precision_gpu", "--debug"],
    ["exe", "-t", "30", "-m", "m.mtx", "-k", "100", "-i", "0", "-r"],
]
opts = []
for argv in argvs:
    arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
    o = RefOpt()
    O.ref().ref_options_parse(len(argv), arr, C.byref(o))
    opts.append({"argv": argv, "matrix_path": o.matrix_path.decode(), "use_sample_matrix": o.use_sample_matrix,
                 "reset": o.reset, "num_tests": o.num_tests, "debug": o.debug,
                 "ignore_matrix_values": o.ignore_matrix_values, "top_k_value": o.top_k_value,
                 "xclbin_path": o.xclbin_path.decode(), "gpu_impl": o.gpu_impl,
                 "use_half_precision_gpu": o.use_half_precision_gpu, "block_size_1d": o.block_size_1d,
                 "block_size_2d": o.block_size_2d, "num_blocks": o.num_blocks})
with open(os.path.join(HERE, "gold_options.json"), "w") as f:
    json.dump(opts, f, indent=1)

# ---- sort_tuples / mean / st_dev -----------------------------------------------------------------------------------
rng = np.random.RandomState(3)
idx = rng.randint(0, 50, 64).astype(np.uint32)
val = np.round(rng.rand(64), 1).astype(np.float32)  # many ties
si, sv = idx.copy(), val.copy()
O.ref().ref_sort_tuples(C.c_ulonglong(64), si.ctypes.data_as(O.u32p), sv.ctypes.data_as(O.f32p))
xs = rng.rand(12).astype(np.float32)
stats = {"idx": idx, "val": val, "sorted_idx": si, "sorted_val": sv, "xs": xs,
         "mean_skip0": np.float32(O.ref().ref_mean(xs.ctypes.data_as(O.f32p), 12, 0)),
         "mean_skip2": np.float32(O.ref().ref_mean(xs.ctypes.data_as(O.f32p), 12, 2)),
         "std_skip0": np.float32(O.ref().ref_st_dev(xs.ctypes.data_as(O.f32p), 12, 0)),
         "std_skip2": np.float32(O.ref().ref_st_dev(xs.ctypes.data_as(O.f32p), 12, 2))}
np.savez_compressed(os.path.join(HERE, "gold_eval.npz"), **stats)
print("golden vectors written to", HERE)
