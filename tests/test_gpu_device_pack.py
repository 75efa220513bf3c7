"""The device packer (csrc/device_pack.hip) against the host packer (csrc/wbscsr.cpp, the role of the
reference's SpMV::packet_coo / packet_coo_partition, src/fpga/src/host_spmv_bscsr.cpp:133-248): identical bytes -- the
packet stream, the packet row table and the partition tables -- for every value type, both packet sizes, matrices with
empty rows, rows longer than a packet, tiny matrices and partition hints that force the capacity loop."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    ra, rb = a.raw(), b.raw()
    assert ra[1] == rb[1], "packet size"
    assert np.array_equal(np.asarray(ra[3]), np.asarray(rb[3])), "part_first"
    assert np.array_equal(np.asarray(ra[4]), np.asarray(rb[4])), "part_count"
    assert np.array_equal(np.asarray(ra[2]), np.asarray(rb[2])), "pkt_row"
    pa, pb = np.asarray(ra[0]), np.asarray(rb[0])
    assert pa.shape == pb.shape
    if not np.array_equal(pa, pb):
        bad = np.flatnonzero(pa != pb)
        raise AssertionError(f"packet stream differs in {bad.size} bytes, first at {bad[0]} (packet {bad[0] // ra[1]}, offset {bad[0] % ra[1]})")
    ia, ib = a.info(), b.info()
    for key in ("nnz", "packed_entries", "packed_bytes", "n_packets", "packet_entries", "n_wave_partitions", "packets_per_partition",
                "precision", "fixed_width"):
        assert ia[key] == ib[key], key


def _coo(pkg, lens, cols, seed=0):
    rng = np.random.RandomState(seed)
    r, c, v = [], [], []
    for i, n in enumerate(lens):
        r += [i] * n
        c += np.sort(rng.randint(0, cols, n)).tolist()
        v += (rng.rand(n) * 1.2).astype(np.float32).tolist()
    return pkg.CooMatrix(len(lens), cols, np.array(r, np.uint32), np.array(c, np.uint32), np.array(v, np.float32))


@pytest.mark.parametrize("precision,width", [("F32", 0), ("F16", 0), ("Q1_7", 0), ("Q1_7_F32", 0), ("FIXED", 20), ("FIXED", 32)])
@pytest.mark.parametrize("rows,cols,nnz,dist,seed,hint", [(20000, 1024, 20, "gamma", 1, 4088), (3000, 512, 40, "uniform", 2, 4088),
                                                          (150000, 300, 25, "gamma", 3, 512), (700, 64, 5, "uniform", 4, 4088),
                                                          (125000, 1024, 20, "gamma", 5, 4064), (40000, 512, 40, "uniform", 6, 1200)])  # (the last two: balanced cuts)
def test_generated_matrices_pack_identically(pkg, precision, width, rows, cols, nnz, dist, seed, hint):
    m = pkg.generate_matrix(rows, cols, nnz, dist, seed)
    kw = dict(k=100, nnz_per_lane=4, n_wave_partitions=hint, precision=getattr(pkg, precision), fixed_width=width)
    _same(pkg.Packed(m, **kw), pkg.Packed(m, on_device=True, **kw))


@pytest.mark.parametrize("C", [4, 8])
@pytest.mark.parametrize("name,lens,cols,hint", [
    ("empty rows and long rows", [0, 3, 0, 0, 700, 1, 0, 256, 257, 0, 5] * 30, 128, 64),
    ("one row", [17], 32, 4088),
    ("one entry", [1], 1, 4088),
    ("leading and trailing empty rows", [0, 0, 0, 4, 9, 0, 0], 16, 8),
    ("rows of exactly one packet", [256] * 40 + [512] * 10, 256, 16),
    ("giant row among small ones", [2] * 500 + [5000] + [2] * 500, 512, 32),
    ("many tiny rows", [1] * 20000, 64, 4088),
])
def test_edge_layouts_pack_identically(pkg, C, name, lens, cols, hint):
    m = _coo(pkg, lens, cols, seed=len(lens))
    kw = dict(k=8, nnz_per_lane=C, n_wave_partitions=hint)
    a, b = pkg.Packed(m, **kw), pkg.Packed(m, on_device=True, **kw)
    _same(a, b)
    r, c, v = b.decode()  # decode(pack(A)) == A through the device packer too
    assert np.array_equal(r, m.row) and np.array_equal(c, m.col) and np.array_equal(v, m.val)


def test_errors_match_the_host_packer(pkg):
    m = _coo(pkg, [3, 4, 5], 16)
    bad = pkg.CooMatrix(3, 16, m.row[::-1].copy(), m.col, m.val)
    with pytest.raises(pkg.TkspmvError) as e:
        pkg.Packed(bad, on_device=True)
    assert e.value.status == pkg._lib.ERR_NOT_SORTED
    bad = pkg.CooMatrix(3, 8, m.row, m.col, m.val)  # column ids up to 15 with 8 columns
    with pytest.raises(pkg.TkspmvError) as e:
        pkg.Packed(bad, on_device=True)
    assert e.value.status == pkg._lib.ERR_INVALID and "column id" in e.value.message
    bad = pkg.CooMatrix(2, 16, m.row, m.col, m.val)  # row id 2 with 2 rows
    with pytest.raises(pkg.TkspmvError) as e:
        pkg.Packed(bad, on_device=True)
    assert e.value.status == pkg._lib.ERR_INVALID and "row id" in e.value.message


def test_large_unsorted_coo_is_refused_without_touching_memory_beyond_the_row_table(pkg):
    """A reversed COO of 200k rows: the row-length table is sized by the LAST entry's row (0 here), so every other row id
    lies beyond it; the count kernel must flag the input as unsorted instead of counting into memory it does not own
    (an engine created before and used after proves that neighbouring allocations are intact)."""
    m = pkg.generate_matrix(20000, 1024, 20, "gamma", 5)
    x = pkg.create_sample_vector(1024, True, False, True, 5)
    eng = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=x, k=100, device=0)
    eng()
    before = eng.read_result()
    big = pkg.generate_matrix(200000, 1024, 20, "gamma", 9)
    bad = pkg.CooMatrix(big.rows, big.cols, big.row[::-1].copy(), big.col[::-1].copy(), big.val[::-1].copy())
    for _ in range(3):
        with pytest.raises(pkg.TkspmvError) as e:
            pkg.Packed(bad, on_device=True)
        assert e.value.status == pkg._lib.ERR_NOT_SORTED
    with pytest.raises(pkg.TkspmvError) as e:  # tkspmv_create's default packer is this one
        pkg.SpMV(bad.row, bad.col, bad.val, bad.rows, bad.cols, vec=x, k=100, device=0)
    assert e.value.status == pkg._lib.ERR_NOT_SORTED
    eng()
    after = eng.read_result()
    assert np.array_equal(before[1], after[1]) and np.array_equal(before[0].view(np.uint32), after[0].view(np.uint32))
    eng.close()


def test_full_size_identical_and_engine_uses_the_device_packer(pkg, oracle, monkeypatch):
    """BASELINE configs[1]'s matrix: identical bytes, the times of both packers, and an engine built either way returns
    the same bits."""
    m = pkg.generate_matrix(1000000, 1024, 20, "gamma", 2)
    n_parts = pkg.Packed.wave_partitions(0)
    t0 = time.perf_counter()
    host = pkg.Packed(m, k=100, nnz_per_lane=4, n_wave_partitions=n_parts)
    t_host = time.perf_counter() - t0
    dev = pkg.Packed(m, k=100, nnz_per_lane=4, n_wave_partitions=n_parts, on_device=True)
    _same(host, dev)
    print(f"\n[packing 1M x 1024, {m.nnz} nnz] host packer {1e3 * t_host:.0f} ms; device packer: upload of the COO "
          f"{dev.pack_ms[0]:.1f} ms + kernels {dev.pack_ms[1]:.1f} ms")
    x = pkg.create_sample_vector(1024, True, False, True, 77)
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("TKSPMV_DEVICE_PACK", flag)
        eng = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=x, k=100, device=0)
        assert eng.info()["pack_on_device"] == int(flag)
        eng()
        res.append(eng.read_result() + (eng.info()["pack_us"],))
        eng.close()
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    print(f"[tkspmv_create, packing step] on the device {res[0][2] / 1e3:.0f} ms, on the host {res[1][2] / 1e3:.0f} ms")
    gi, gv = oracle.gold_topk(m.row, m.col, m.val, x, 100)
    assert set(res[0][1].tolist()) == set(gi.tolist())


# ---- the wave-sliced ELL layout of the multi-query kernel (wsell.cpp / device_pack.hip: sell_scatter_kernel) -------------
@pytest.mark.parametrize("precision", ["F32", "Q1_7_F32"])
@pytest.mark.parametrize("rows,cols,nnz,dist,seed,hint", [(20000, 1024, 20, "gamma", 1, 4088), (3000, 512, 40, "uniform", 2, 4088),
                                                          (150000, 300, 25, "gamma", 3, 512), (700, 64, 5, "uniform", 4, 4088)])
def test_sell_layout_packs_identically(pkg, precision, rows, cols, nnz, dist, seed, hint):
    m = pkg.generate_matrix(rows, cols, nnz, dist, seed)
    r = pkg.sell_pack_device_check(m, hint, precision=getattr(pkg, precision))
    # (byte chunks of at most 1022 columns carry 12-bit column words: 640 bytes instead of 768)
    assert r["identical"] and r["chunks"] > 0 and r["stream_bytes"] == r["chunks"] * (1536 if precision == "F32" else (640 if cols <= 1022 else 768))


@pytest.mark.parametrize("precision", ["F32", "Q1_7_F32"])
@pytest.mark.parametrize("name,lens,cols,hint", [
    ("empty rows and rows of several lanes", [0, 3, 0, 0, 700, 1, 0, 256, 257, 0, 5, 64, 65, 80] * 30, 128, 64),
    ("one row", [17], 32, 4088),
    ("one entry", [1], 1, 4088),
    ("leading and trailing empty rows", [0, 0, 0, 4, 9, 0, 0], 16, 8),
    ("a row of 4096 entries (64 lanes) among small ones", [2] * 500 + [4096] + [2] * 500, 512, 32),
    ("many tiny rows", [1] * 20000, 64, 4088),
    ("more partitions than slices", [5] * 100, 64, 4088),
])
def test_sell_edge_layouts_pack_identically(pkg, precision, name, lens, cols, hint):
    m = _coo(pkg, lens, cols, seed=len(lens))
    assert pkg.sell_pack_device_check(m, hint, precision=getattr(pkg, precision))["identical"]


def test_sell_errors_match_the_host_packer(pkg):
    m = _coo(pkg, [3, 4, 5], 16)
    for bad, what in ((pkg.CooMatrix(3, 16, m.row[::-1].copy(), m.col, m.val), "not sorted"),
                      (pkg.CooMatrix(3, 8, m.row, m.col, m.val), "column id"),
                      (pkg.CooMatrix(2, 16, m.row, m.col, m.val), "row id"),
                      (_coo(pkg, [2, 5000, 2], 512), "too long")):
        with pytest.raises(pkg.TkspmvError) as e:
            pkg.sell_pack_device_check(bad, 64)
        assert e.value.status == pkg._lib.ERR_INVALID and what in e.value.message


def test_sell_full_size_identical_and_multi_query_engine_uses_it(pkg, oracle, monkeypatch):
    """BASELINE configs[1]'s matrix in the row-per-lane layout: identical bytes and both packers' times; a multi-query
    engine built either way returns the same bits."""
    m = pkg.generate_matrix(1000000, 1024, 20, "gamma", 2)
    r = pkg.sell_pack_device_check(m, 4088)
    assert r["identical"]
    print(f"\n[row-per-lane layout, 1M x 1024, {r['stream_bytes'] / 1e6:.0f} MB] host packer {r['host_ms']:.0f} ms; device packer: plan "
          f"{r['plan_ms']:.0f} ms + uploads (COO included) {r['upload_ms']:.0f} ms + fill kernel {r['fill_ms']:.1f} ms")
    import torch
    rng = np.random.RandomState(5)
    X = rng.rand(8, 1024).astype(np.float32)
    dxs = torch.from_numpy(X).cuda()
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("TKSPMV_DEVICE_PACK", flag)
        eng = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=X[0], k=100, device=0, multi_q=4)
        assert eng.info()["pack_on_device"] == int(flag) and eng.info()["multi_q"] == 4
        out_i = torch.full((8, 100), -1, dtype=torch.int32, device="cuda")
        out_v = torch.full((8, 100), -1.0, dtype=torch.float32, device="cuda")
        eng.enqueue_multi(dxs.data_ptr(), 8, out_i.data_ptr(), out_v.data_ptr())
        eng.synchronize()
        res.append((out_i.cpu().numpy(), out_v.cpu().numpy(), eng.info()["multi_pack_us"]))
        eng.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
    print(f"[tkspmv_create, packing of the row-per-lane copy] device path {res[0][2] / 1e3:.0f} ms, host path {res[1][2] / 1e3:.0f} ms")
    gi, gv = oracle.gold_topk(m.row, m.col, m.val, X[3], 100)
    assert set(res[0][0][3].tolist()) == set(gi.tolist())
