"""The C-ABI shared library: loads, exports every symbol include/tkspmv.h declares, and fails loudly (no CPU
fallback) when there is no GPU. No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tkspmv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tkspmv_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} is declared in include/tkspmv.h but not exported"
    assert sorted(pkg._lib.EXPORTED_SYMBOLS) == declared


def test_struct_layouts_match_header(pkg):
    """Sizes the C compiler sees for the ABI structs == the ctypes mirrors."""
    src = r'''
    #include <stdio.h>
    #include "tkspmv.h"
    int main(void) { printf("%zu %zu %zu %zu %zu\n", sizeof(tkspmv_desc), sizeof(tkspmv_info), sizeof(tkspmv_timing),
                            sizeof(tkspmv_coo), sizeof(tkspmv_options)); return 0; }
    '''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    L = pkg._lib
    assert [int(x) for x in out] == [C.sizeof(L.Desc), C.sizeof(L.Info), C.sizeof(L.Timing), C.sizeof(L.Coo),
                                    C.sizeof(L.OptionsC)]


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    m = pkg.generate_matrix(100, 32, 5, "uniform", 1)
    with pytest.raises(pkg.TkspmvError) as e:
        pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=8)
    assert e.value.status == pkg._lib.ERR_DEVICE
    assert "no CPU fallback" in e.value.message


def test_argument_validation_before_device(pkg):
    """Descriptor errors are reported with ERR_INVALID / ERR_UNSUPPORTED regardless of the device."""
    m = pkg.generate_matrix(50, 16, 4, "uniform", 2)
    for kw, status in ((dict(k=0), pkg._lib.ERR_INVALID), (dict(k=2000), pkg._lib.ERR_INVALID),
                       (dict(k=8, precision=pkg.Q1_7, nnz_per_lane=8), pkg._lib.ERR_UNSUPPORTED),
                       (dict(k=8, precision=7), pkg._lib.ERR_INVALID),
                       (dict(k=8, impl=9), pkg._lib.ERR_INVALID),
                       (dict(k=100, partitions=70000, k_per_partition=8), pkg._lib.ERR_INVALID)):
        with pytest.raises(pkg.TkspmvError) as e:
            pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, **kw)
        assert e.value.status == status, kw


def test_executable_reports_errors_like_the_reference(pkg, tmp_path):
    exe = os.path.join(ROOT, "bin", "approximate-spmv-mi355x-topk")
    assert os.path.exists(exe), "run make"
    r = subprocess.run([exe, "-m", str(tmp_path / "nope.mtx")], capture_output=True, text=True)
    assert r.returncode == 1 and "not found" in r.stderr  # utils.hpp:486-490
    bad = tmp_path / "bad.mtx"
    bad.write_text("garbage\n")
    r = subprocess.run([exe, "-m", str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "Could not process Matrix Market banner" in r.stdout  # utils.hpp:493-496
    if pkg.device_count() == 0:
        g = pkg.generate_matrix(200, 64, 6, "gamma", 1)
        p = tmp_path / "ok.mtx"
        pkg.write_mtx(str(p), g, index_base=1)
        # a one-based file (generator output) read zero-based, the reference's compiled-in behaviour and the default here:
        # the reference then reads x out of bounds; this executable says what is wrong
        r = subprocess.run([exe, "-m", str(p), "-k", "8", "-t", "1"], capture_output=True, text=True)
        assert r.returncode == 1 and "TKSPMV_INDEX_BASE=1" in r.stderr
        r = subprocess.run([exe, "-m", str(p), "-k", "8", "-t", "1"], capture_output=True, text=True,
                           env=dict(os.environ, TKSPMV_INDEX_BASE="1"))
        assert r.returncode == 1 and "no HIP device" in r.stderr


def test_reference_side_host_program_builds_against_the_reference_headers(pkg):
    """oracle/ref_host_mi355x.cpp -- `struct SpMV` bound to the C ABI inside a main() written with
    the reference's own Options / readMtx / coo_t / create_sample_vector / spmv_coo_gold_top_k / sort_tuples -- compiles with
    plain g++ against the headers where they lie under /root/reference (`make ref`) and links libtkspmv.so. Without a GPU the
    program stops where the engine is created, loudly."""
    if not os.path.isdir("/root/reference"):
        pytest.skip("the reference tree is only present in the build container")
    r = subprocess.run(["make", "-C", ROOT, "ref"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    exe = os.path.join(ROOT, "oracle", "_ref", "host_spmv_topk_mi355x")
    assert os.path.exists(exe)
    if pkg.device_count() == 0:
        r = subprocess.run([exe, "-m", os.path.join(ROOT, "tests", "golden", "small_0indexed.mtx"), "-k", "8", "-t", "1"],
                           capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stderr


def test_options_are_one_documented_table_settable_through_the_abi(pkg, monkeypatch):
    """tkspmv_set_option / tkspmv_get_option / tkspmv_option_info (include/tkspmv.h): every switch the library reads is a row of
    csrc/options.cpp's table; the environment is the same option for shell-driven runs and a value set through the call wins; a
    name outside the table is refused; and no library source calls getenv for an engine option anywhere else."""
    import glob
    import os
    import re
    opts = {o["name"]: o for o in pkg.options()}
    assert {"LOCAL", "BATCH", "SELECTORS", "BAR_X", "DIST_NO_NCCL", "TRACE"} <= set(opts)
    assert all(o["doc"] and o["values"] and o["kind"] in ("behaviour", "layout", "tuning", "diagnostic") for o in opts.values())
    monkeypatch.delenv("TKSPMV_SELECTORS", raising=False)
    assert pkg.get_option("SELECTORS") is None
    monkeypatch.setenv("TKSPMV_SELECTORS", "2")
    assert pkg.get_option("SELECTORS") == "2"
    pkg.set_option("SELECTORS", 3)
    assert pkg.get_option("SELECTORS") == "3"
    pkg.set_option("SELECTORS", None)
    assert pkg.get_option("SELECTORS") == "2"
    with pytest.raises(pkg.TkspmvError):
        pkg.set_option("NO_SUCH_SWITCH", 1)
    assert pkg.get_option("NO_SUCH_SWITCH") is None
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "approximate-spmv-topk_amd", "csrc")
    used = set()
    for path in glob.glob(os.path.join(csrc, "**", "*.*"), recursive=True):
        src = open(path).read()
        if os.path.basename(path) not in ("options.cpp", "main_topk.cpp"):  # (main_topk.cpp: the host PROGRAM's own four settings)
            assert "getenv" not in src, f"{path} reads the environment behind the option table's back"
        used |= set(re.findall(r'\bopt(?:_set|_int)?\("([A-Z0-9_]+)"', src))
    assert used == set(opts), (used - set(opts), set(opts) - used)  # no undocumented switch, no documented ghost
