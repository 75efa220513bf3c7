"""Host-side mirror of the reference's common layer (Options, readMtx, create_sample_vector, evaluation helpers,
generator) and the wave-BSCSR packer, checked against golden vectors produced by the reference itself."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_options_match_reference(pkg):
    with open(os.path.join(GOLD, "gold_options.json")) as f:
        cases = json.load(f)
    for c in cases:
        o = pkg.Options.parse(c["argv"])
        for key in ("matrix_path", "xclbin_path", "num_tests", "debug", "top_k_value", "gpu_impl", "block_size_1d",
                    "block_size_2d", "num_blocks"):
            assert getattr(o, key) == c[key], (c["argv"], key)
        for key in ("use_sample_matrix", "reset", "ignore_matrix_values", "use_half_precision_gpu"):
            assert bool(getattr(o, key)) == bool(c[key]), (c["argv"], key)
    # the -r quirk: "--no_reset" leaves reset on (options.hpp:92-94)
    assert pkg.Options.parse(["exe", "-r"]).reset is True


def test_read_mtx_matches_reference(pkg):
    z = np.load(os.path.join(GOLD, "gold_read_mtx.npz"))
    p1, p0 = os.path.join(GOLD, "small_1indexed.mtx"), os.path.join(GOLD, "small_0indexed.mtx")
    for tag, path, base in (("one_as_one", p1, 1), ("zero_as_zero", p0, 0), ("one_as_zero", p1, 0)):
        m = pkg.read_mtx(path, index_base=base)
        hdr = z[f"{tag}_hdr"]
        assert (m.rows, m.cols, m.nnz) == (int(hdr[0]), int(hdr[1]), int(hdr[2]))
        assert np.array_equal(m.row, z[f"{tag}_row"]) and np.array_equal(m.col, z[f"{tag}_col"])
        assert np.array_equal(m.val.view(np.uint32), z[f"{tag}_val"].view(np.uint32))
        assert m.num_rows_coo == int(z[f"{tag}_num_rows_coo"][0])
    # auto detection picks the right base for both files and yields identical matrices
    a, b = pkg.read_mtx(p1, index_base=-1), pkg.read_mtx(p0, index_base=-1)
    assert (a.index_base, b.index_base) == (1, 0)
    assert np.array_equal(a.row, b.row) and np.array_equal(a.col, b.col) and np.array_equal(a.val, b.val)


def test_read_mtx_errors(pkg, tmp_path):
    with pytest.raises(pkg.TkspmvError) as e:
        pkg.read_mtx(str(tmp_path / "missing.mtx"))
    assert e.value.status == pkg._lib.ERR_IO and "not found" in e.value.message
    bad = tmp_path / "bad.mtx"
    bad.write_text("this is not a banner\n1 1 1\n0 0 1.0\n")
    with pytest.raises(pkg.TkspmvError) as e:
        pkg.read_mtx(str(bad))
    assert "Could not process Matrix Market banner" in e.value.message
    short = tmp_path / "short.mtx"
    short.write_text("%%MatrixMarket matrix coordinate real general\n%\n3 3 4\n0 0 1.0\n1 1 2.0\n")
    with pytest.raises(pkg.TkspmvError) as e:
        pkg.read_mtx(str(short))
    assert "Not enough rows" in e.value.message
    # comments, blank-ish size line handling, scientific notation, pattern files, symmetric banner
    ok = tmp_path / "ok.mtx"
    ok.write_text("%%MatrixMarket matrix coordinate real general\n% c1\n% c2\n3 4 3\n0 1 1e-3\n1 3 2.5E+0\n2 0 .5\n")
    m = pkg.read_mtx(str(ok))
    assert (m.rows, m.cols, m.nnz) == (3, 4, 3) and np.allclose(m.val, [1e-3, 2.5, 0.5])
    pat = tmp_path / "pat.mtx"
    pat.write_text("%%MatrixMarket matrix coordinate pattern general\n2 2 2\n0 1\n1 0\n")
    m = pkg.read_mtx(str(pat))
    assert np.array_equal(m.val, [1.0, 1.0])
    sym = tmp_path / "sym.mtx"
    sym.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 3\n0 0 1.0\n1 0 2.0\n2 1 3.0\n")
    m = pkg.read_mtx(str(sym))
    assert m.symmetric and m.nnz == 5 and np.all(np.diff(m.row.astype(np.int64)) >= 0)
    m = pkg.read_mtx(str(ok), read_values=False)
    assert np.array_equal(m.val, [1.0, 1.0, 1.0])


def test_sample_vector_matches_reference(pkg):
    z = np.load(os.path.join(GOLD, "gold_sample_vector.npz"))
    for size in (16, 1024):
        for sd in (1, 7, 123):
            assert np.array_equal(pkg.create_sample_vector(size, True, False, True, sd), z[f"norm_{size}_{sd}"])
            assert np.array_equal(pkg.create_sample_vector(size, True, True, False, sd), z[f"sum_{size}_{sd}"])
    assert np.array_equal(pkg.create_sample_vector(8, False, True, False, 0), z["ones_sum_8"])
    a, b = pkg.create_sample_vector(64, True, False, True, 0), pkg.create_sample_vector(64, True, False, True, 0)
    assert not np.array_equal(a, b)  # seed 0 = random_device
    assert abs(np.linalg.norm(a.astype(np.float64)) - 1.0) < 1e-6


def test_generator_distributions(pkg):
    """create_matrices.py:83-104: gamma(3, avg/3) truncated, at least 1; uniform in [avg//2, int(1.5*avg)];
    columns drawn with replacement and sorted; values L2-normalised per row."""
    for dist, lo, hi in (("gamma", 1, 10 ** 6), ("uniform", 10, 30)):
        m = pkg.generate_matrix(20000, 1024, 20, dist, 5)
        deg = np.bincount(m.row, minlength=m.rows)
        assert deg.min() >= lo and deg.max() <= hi
        assert abs(deg.mean() - (19.5 if dist == "gamma" else 20.0)) < 0.4
        assert np.all(np.diff(m.row.astype(np.int64)) >= 0)
        same_row = np.diff(m.row.astype(np.int64)) == 0
        assert np.all(np.diff(m.col.astype(np.int64))[same_row] >= 0)
        assert (np.diff(m.col.astype(np.int64))[same_row] == 0).sum() > 0  # duplicates exist, and are kept
        n2 = np.bincount(m.row, weights=m.val.astype(np.float64) ** 2, minlength=m.rows)
        assert np.allclose(n2, 1.0, atol=1e-5)
        assert m.val.min() >= 0 and m.col.max() < 1024
    a, b = pkg.generate_matrix(500, 64, 8, "gamma", 9), pkg.generate_matrix(500, 64, 8, "gamma", 9)
    assert np.array_equal(a.row, b.row) and np.array_equal(a.col, b.col) and np.array_equal(a.val, b.val)


def test_mtx_round_trip(pkg, tmp_path):
    m = pkg.generate_matrix(300, 100, 7, "uniform", 4)
    for base in (0, 1):
        p = tmp_path / f"m{base}.mtx"
        pkg.write_mtx(str(p), m, index_base=base)
        assert p.read_text().startswith("%%MatrixMarket matrix coordinate real general\n%\n300 100 ")
        r = pkg.read_mtx(str(p), index_base=base)
        assert np.array_equal(r.row, m.row) and np.array_equal(r.col, m.col)
        assert np.allclose(r.val, m.val, rtol=1e-9)  # 10 significant digits


def _special_matrix(rows, cols, lens, seed=0):
    rng = np.random.RandomState(seed)
    r, c, v = [], [], []
    for i, n in enumerate(lens):
        cs = np.sort(rng.randint(0, cols, n))
        r += [i] * n
        c += cs.tolist()
        v += rng.rand(n).astype(np.float32).tolist()
    return rows, cols, np.array(r, np.uint32), np.array(c, np.uint32), np.array(v, np.float32)


@pytest.mark.parametrize("C", [4, 8])
@pytest.mark.parametrize("parts", [1, 7, 64, 4096])
def test_pack_decode_round_trip(pkg, C, parts):
    lens = [1, 1, 0, 3, 171, 0, 0, 300, 1, 2, 1100, 5, 0, 1, 64, 256, 255, 257, 1, 1] * 3 + [4]
    rows, cols, r, c, v = _special_matrix(len(lens) + 5, 1024, lens, 1)  # 5 trailing empty rows
    m = pkg.CooMatrix(rows, cols, r, c, v)
    p = pkg.Packed(m, nnz_per_lane=C, n_wave_partitions=parts)
    dr, dc, dv = p.decode()
    assert np.array_equal(dr, r) and np.array_equal(dc, c) and np.array_equal(dv, v)
    info = p.info()
    assert info["packet_entries"] == 64 * C and info["n_wave_partitions"] <= parts
    assert info["packed_entries"] == info["n_packets"] * 64 * C
    packets, packet_bytes, pkt_row, part_first, part_count = p.raw()
    # (fp32 over at most 1024 columns, 4 entries per lane: 12-bit column words, 5.5 bytes per entry)
    assert packet_bytes == (64 * C * 11 // 2 if C == 4 else 64 * C * 6) and len(packets) == info["n_packets"] * packet_bytes
    assert part_count.sum() == info["n_packets"] and np.all(part_first[1:] == np.cumsum(part_count)[:-1])
    g = pkg.generate_matrix(3000, 512, 20, "gamma", 3)
    p = pkg.Packed(g, nnz_per_lane=C, n_wave_partitions=parts)
    dr, dc, dv = p.decode()
    assert np.array_equal(dr, g.row) and np.array_equal(dc, g.col) and np.array_equal(dv, g.val)
    # padding stays small: at most one packet per partition
    assert p.info()["packed_entries"] - g.nnz <= p.info()["n_wave_partitions"] * 64 * C


@pytest.mark.parametrize("rows,hint", [(125000, 4064), (30000, 1016), (47000, 1504)])
def test_balanced_cuts_fill_the_waves_asked_for_and_lose_nothing(pkg, monkeypatch, rows, hint):
    """Round 5 (wbscsr.cpp, fill_partitions_balanced): partitions of equal capacity fill 79 % of the waves of a 125k-row shard (3229
    partitions of 3 packets for 4064 waves: a batch launch's workgroups stream 6 or 7 each); where that share is under 7/8 and a
    partition holds two packets or more, the packets are dealt out over the count asked for, floor or ceil of the mean each. The
    stream decodes to the matrix either way, every partition starts on a row boundary, and the tables stay consistent."""
    m = pkg.generate_matrix(rows, 1024, 20, "gamma", 2)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("TKSPMV_BALANCED_CUTS", flag)
        p = pkg.Packed(m, nnz_per_lane=4, n_wave_partitions=hint)
        dr, dc, dv = p.decode()
        assert np.array_equal(dr, m.row) and np.array_equal(dc, m.col) and np.array_equal(dv, m.val)
        packets, packet_bytes, pkt_row, part_first, part_count = p.raw()
        assert part_count.sum() == p.info()["n_packets"] and np.all(part_first[1:] == np.cumsum(part_count)[:-1])
        assert np.all(np.diff(pkt_row.astype(np.int64)) >= 0) and len(part_count) <= hint
        res[flag] = part_count
    uniform, balanced = res["0"], res["1"]
    assert len(uniform) * 8 < hint * 7, "the case is meant to be one the uniform cut leaves waves idle in"
    assert len(balanced) > len(uniform) and len(balanced) * 16 >= hint * 15  # (the waves asked for, or a handful fewer)
    assert balanced.max() - balanced.min() <= 2 and balanced.min() >= 1  # (floor / ceil of the mean; a long row may add a packet)
    # a batch launch gives wave w of workgroup b partition w * n_wg + b: the packets per workgroup are within a few of each other
    n_wg = hint // 8
    per_wg = np.array([balanced[b::n_wg].sum() for b in range(n_wg)])
    per_wg_u = np.array([uniform[b::n_wg].sum() for b in range(n_wg)])
    assert per_wg.max() - per_wg.min() <= 8 and per_wg.max() < per_wg_u.max()


def test_shortest_partition_depends_on_the_matrix_size_alone(pkg, monkeypatch):
    """wbscsr.hpp: min_packets_per_partition_for -- every packer (host, device, tkspmv_pack, the engine) must cut the same
    partitions from the same hint. Small matrices (the shards of a strong-scaled run) get partitions from one or two packets
    up, so that every streaming wave of a 256-CU launch has something to stream; large ones at least four packets each."""
    tiny = pkg.generate_matrix(40000, 1024, 20, "gamma", 3)      # ~3100 packets of 256 entries: up to 1 per partition
    small = pkg.generate_matrix(150000, 1024, 20, "gamma", 3)    # ~11700 packets: 2+
    large = pkg.generate_matrix(800000, 1024, 20, "gamma", 3)    # ~62500 packets: 4+
    for m, lo, hi in ((tiny, 1, 2), (small, 2, 3), (large, 4, 1000)):
        i = pkg.Packed(m, nnz_per_lane=4, n_wave_partitions=16384).info()
        assert lo <= i["packets_per_partition"] <= hi, (m.rows, i["packets_per_partition"])
        assert i["n_wave_partitions"] * i["packets_per_partition"] >= i["n_packets"]
    # the same hint on the same matrix gives the same cut (what the parity tests rely on when they re-pack on the host)
    a = pkg.Packed(small, nnz_per_lane=4, n_wave_partitions=4064).raw()
    b = pkg.Packed(small, nnz_per_lane=4, n_wave_partitions=4064).raw()
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    monkeypatch.setenv("TKSPMV_SMALL_PACKETS", "0")  # round 2's rule: four packets at least, whatever the size
    assert pkg.Packed(tiny, nnz_per_lane=4, n_wave_partitions=16384).info()["packets_per_partition"] >= 4


def test_fp32_column_words_travel_as_12_bits(pkg, oracle, monkeypatch):
    """TKSPMV_F32 over at most 1024 columns packs its column words (10 bits of column, 2 flags) into 12 bits (the default;
    TKSPMV_F32_C12=0 keeps 16): 1408-byte packets instead of 1536. Same entries, same order; the bits are where csrc/wbscsr.hpp says (a split plane: one
    dword and one halfword per lane); the order-matched oracle computes the same scores, bit for bit, from either layout."""
    g = pkg.generate_matrix(5000, 1024, 20, "gamma", 9)
    p12 = pkg.Packed(g, nnz_per_lane=4, n_wave_partitions=64)
    monkeypatch.setenv("TKSPMV_F32_C12", "0")
    p16 = pkg.Packed(g, nnz_per_lane=4, n_wave_partitions=64)
    monkeypatch.delenv("TKSPMV_F32_C12")
    r12, r16 = p12.raw(), p16.raw()
    assert r12[1] == 1408 and r16[1] == 1536
    for a, b in zip(r12[2:], r16[2:]):
        assert np.array_equal(np.asarray(a), np.asarray(b))  # packet rows and partitions do not depend on the word size
    for p in (p12, p16):
        dr, dc, dv = p.decode()
        assert np.array_equal(dr, g.row) and np.array_equal(dc, g.col) and np.array_equal(dv, g.val)
    a12 = np.asarray(r12[0]).reshape(-1, 1408)
    a16 = np.asarray(r16[0]).reshape(-1, 1536)
    assert np.array_equal(a12[:, :1024], a16[:, :1024])  # the values
    # the split plane, rebuilt here from the 16-bit words (column << 2 | SKIP << 1 | ROW_END), lane l owning entries 4l..4l+3:
    #   A[l] = col0 << 2 | col1 << 12 | col2 << 22 | SKIP0 | SKIP1 << 1
    #   B[l] = col3 << 2 | SKIP2 | SKIP3 << 1 | ROW_END0..3 << 12
    cw16 = a16[:, 1024:].copy().view(np.uint16).astype(np.uint32).reshape(-1, 64, 4)  # [packets][lane][j]
    col, skip, end = cw16 >> 2, (cw16 >> 1) & 1, cw16 & 1
    A = (col[:, :, 0] << 2) | (col[:, :, 1] << 12) | (col[:, :, 2] << 22) | skip[:, :, 0] | (skip[:, :, 1] << 1)
    B = (col[:, :, 3] << 2) | skip[:, :, 2] | (skip[:, :, 3] << 1) | (end[:, :, 0] << 12) | (end[:, :, 1] << 13) | (end[:, :, 2] << 14) | (end[:, :, 3] << 15)
    # ... laid out per PAIR of lanes as 12 bytes [A_even][B_even | B_odd << 16][A_odd] (one dwordx2 per lane at a 4-byte boundary)
    blocks = np.stack([A[:, 0::2], B[:, 0::2] | (B[:, 1::2] << 16), A[:, 1::2]], axis=2)  # [packets][pair][3 dwords]
    plane = blocks.astype("<u4").view(np.uint8).reshape(-1, 384)
    assert np.array_equal(plane, a12[:, 1024:])
    x = pkg.create_sample_vector(1024, True, False, True, 5)
    y12, pr12 = oracle.packed_scores(r12, x, g.rows, 4)
    y16, pr16 = oracle.packed_scores(r16, x, g.rows, 4)
    assert np.array_equal(y12.view(np.uint32), y16.view(np.uint32)) and np.array_equal(pr12, pr16)
    # more than 1024 columns, or 8 entries per lane: 16-bit words as before
    assert pkg.Packed(pkg.generate_matrix(2000, 2048, 10, "uniform", 1), nnz_per_lane=4).raw()[1] == 1536
    assert pkg.Packed(g, nnz_per_lane=8).raw()[1] == 3072


def test_pack_rejects_bad_input(pkg):
    rows, cols, r, c, v = _special_matrix(10, 16, [2] * 10)
    bad = r.copy()
    bad[3], bad[4] = bad[4], bad[3] + 5
    with pytest.raises(pkg.TkspmvError) as e:
        pkg.Packed(pkg.CooMatrix(rows, cols, np.array([3, 1, 2], np.uint32), np.array([0, 1, 2], np.uint32),
                                 np.ones(3, np.float32)))
    assert e.value.status == pkg._lib.ERR_NOT_SORTED
    with pytest.raises(pkg.TkspmvError) as e:
        pkg.Packed(pkg.CooMatrix(rows, cols, r, np.full_like(c, 16), v))
    assert e.value.status == pkg._lib.ERR_INVALID
    with pytest.raises(pkg.TkspmvError):
        pkg.Packed(pkg.CooMatrix(5, cols, r, c, v))  # row ids >= rows
    with pytest.raises(pkg.TkspmvError):
        pkg.Packed(pkg.CooMatrix(rows, 20000, r, c, v))  # more than 16384 columns
    empty = pkg.Packed(pkg.CooMatrix(4, 8, np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32)))
    assert empty.info()["n_packets"] == 0 and len(empty.decode()[0]) == 0


def test_eval_helpers_match_reference(pkg):
    """sort_tuples / mean / st_dev mirrors live in the C++ host layer; exercised through the executable in the GPU
    tests. Here: the Python-visible order contract of read_result (value desc, index desc) on the golden ties."""
    z = np.load(os.path.join(GOLD, "gold_eval.npz"))
    order = np.lexsort((-z["idx"].astype(np.int64), -z["val"].astype(np.float64)))
    assert np.array_equal(z["idx"][order], z["sorted_idx"]) and np.array_equal(z["val"][order], z["sorted_val"])


# ---- packed-matrix cache (.tkspmv files) -----------------------------------------------------------------
@pytest.mark.parametrize("precision", ["F32", "Q1_7", "F16"])
def test_packed_file_round_trip_is_bit_identical(pkg, tmp_path, precision):
    m = pkg.generate_matrix(3000, 512, 20, "gamma", 9)
    p = pkg.Packed(m, k=50, n_wave_partitions=64, precision=getattr(pkg, precision))
    path = tmp_path / "m.tkspmv"
    p.save(path)
    q = pkg.Packed.load(path)
    a, b = p.raw(), q.raw()
    assert a[1] == b[1]
    for u, v in zip((a[0], a[2], a[3], a[4]), (b[0], b[2], b[3], b[4])):
        assert np.array_equal(u, v)
    ia, ib = p.info(), q.info()
    for key in ("rows", "cols", "nnz", "packed_entries", "packed_bytes", "n_packets", "packet_entries",
                "n_wave_partitions", "packets_per_partition", "precision"):
        assert ia[key] == ib[key], key
    ra, ca, va = p.decode()
    rb, cb, vb = q.decode()
    assert np.array_equal(ra, rb) and np.array_equal(ca, cb) and np.array_equal(va, vb)


# ---- generic fixed point (TKSPMV_FIXED: the FPGA's real_type for any FIXED_WIDTH) ---------------------------------------
@pytest.mark.parametrize("width", [8, 20, 21, 25, 26, 32])
def test_fixed_point_values_are_truncated_to_the_width(pkg, tmp_path, width):
    """ap_ufixed<W,1,AP_TRN_ZERO> (fpga_types.hpp:20): W-1 fraction bits, truncation toward zero, values in [0, 2);
    the packer saturates above the range. decode(pack(A)) gives floor(v * 2^(W-1)) / 2^(W-1); the width survives a
    .tkspmv round trip."""
    rng = np.random.RandomState(width)
    n = 4000
    vals = np.concatenate([rng.rand(n - 6).astype(np.float32) * np.float32(1.999),
                           np.array([0.0, 1.0, 1.5, 2.0, 3.7, 1e-9], dtype=np.float32)])
    m = pkg.CooMatrix(n, 8, np.arange(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32), vals)
    p = pkg.Packed(m, n_wave_partitions=16, precision=pkg.FIXED, fixed_width=width)
    info = p.info()
    assert info["precision"] == pkg.FIXED and info["fixed_width"] == width
    r, c, v = p.decode()
    scale = 2.0 ** (width - 1)
    want = np.minimum(np.floor(vals.astype(np.float64) * scale), 2.0 ** width - 1) / scale
    assert np.array_equal(r, m.row)
    assert np.array_equal(v, want.astype(np.float32))  # (u32 -> fp32 rounds to nearest even; so does this cast)
    path = tmp_path / "f.tkspmv"
    p.save(path)
    q = pkg.Packed.load(path)
    assert q.info()["fixed_width"] == width and q.info()["precision"] == pkg.FIXED
    assert np.array_equal(q.raw()[0], p.raw()[0])


def test_fixed_width_is_validated(pkg):
    m = pkg.generate_matrix(100, 64, 5, "uniform", 1)
    for kw in (dict(precision=pkg.FIXED, fixed_width=7), dict(precision=pkg.FIXED, fixed_width=33),
               dict(precision=pkg.F32, fixed_width=20), dict(precision=pkg.Q1_7, fixed_width=8)):
        with pytest.raises(pkg.TkspmvError) as ei:
            pkg.Packed(m, n_wave_partitions=4, **kw)
        assert ei.value.status == pkg._lib.ERR_INVALID, kw


def test_fixed_point_model_known_answers(oracle):
    """The integer model of real_type on hand-computed cases: exact products, truncation of a product, wrap of a
    product at 2.0, wrap of a sum at 2.0, saturation of an input -- and width 8 equals the Q1.7 model."""
    row = np.array([0, 1, 2, 2, 3, 4], dtype=np.uint32)
    col = np.array([0, 1, 0, 0, 2, 3], dtype=np.uint32)
    val = np.array([0.5, 1.5, 1.0, 1.0, 0.75, 5.0], dtype=np.float32)
    x = np.array([1.0, 1.5, 2.0 ** -19, 0.5], dtype=np.float32)
    y, present = oracle.fixed_scores(row, col, val, x, 5, 20)
    assert present.tolist() == [1, 1, 1, 1, 1]
    assert y[0] == 0.5              # 0.5 * 1.0
    assert y[1] == 0.25             # 1.5 * 1.5 = 2.25 wraps at 2.0
    assert y[2] == 0.0              # 1.0 + 1.0 wraps at 2.0
    assert y[3] == 0.0              # 0.75 * 2^-19 = 1.5 * 2^-20 is below one unit of 2^-19... truncated to 0
    assert y[4] == np.float32((2.0 ** 20 - 1) / 2.0 ** 19 * 0.5 - 2.0 ** -19 * 0.5)  # 5.0 saturates at 2 - 2^-19; product truncated
    y25, _ = oracle.fixed_scores(row, col, val, x, 5, 25)
    assert y25[3] == np.float32(0.75 * 2.0 ** -19)  # representable with 24 fraction bits
    rng = np.random.RandomState(5)
    n = 20000
    r = np.sort(rng.randint(0, 900, n)).astype(np.uint32)
    c = rng.randint(0, 64, n).astype(np.uint32)
    v = (rng.rand(n) * 2.2).astype(np.float32)
    xx = (rng.rand(64) * 1.9).astype(np.float32)
    y8, p8 = oracle.fixed_scores(r, c, v, xx, 900, 8)
    yq, pq = oracle.q17_scores(r, c, v, xx, 900)
    assert np.array_equal(y8, yq) and np.array_equal(p8, pq)


# ---- wave-sliced ELL layout of the multi-query kernel (wsell.hpp) ------------------------------------------------------
@pytest.mark.parametrize("rows,cols,nnz,dist,parts", [(5000, 1024, 20, "gamma", 64), (700, 64, 6, "uniform", 4088),
                                                       (64, 16, 3, "uniform", 1), (130, 1024, 40, "gamma", 3)])
def test_sliced_ell_round_trip(pkg, rows, cols, nnz, dist, parts):
    """decode(pack(A)) gives every row back with its entries in their original order (duplicates kept, padding dropped),
    rows sorted by length inside the stream; partitions are balanced; padding stays small."""
    m = pkg.generate_matrix(rows, cols, nnz, dist, 4)
    r, c, v, info = pkg.sell_roundtrip(m, parts)
    assert r.shape[0] == m.nnz
    order = np.argsort(r, kind="stable")  # back to row-major; stable keeps the order inside a row
    assert np.array_equal(r[order], m.row) and np.array_equal(c[order], m.col)
    assert np.array_equal(v[order].view(np.uint32), m.val.view(np.uint32))
    lens = np.bincount(m.row)
    n_lanes = int(np.ceil(lens[lens > 0] / 64).sum())  # a row of more than 64 entries takes several lanes
    assert (n_lanes + 63) // 64 <= info["slices"] <= (n_lanes + 63) // 64 + 2 and info["partitions"] == min(parts, info["slices"])
    assert info["padded_entries"] == info["chunks"] * 256 and info["stream_bytes"] == info["chunks"] * 1536
    if rows >= 5000:
        assert info["padded_entries"] < 1.15 * m.nnz  # sorted rows: what is lost is mostly the rounding to 4 entries
        assert info["most_chunks_per_partition"] <= info["chunks"] / info["partitions"] + 16  # within one slice of the mean


@pytest.mark.parametrize("cols,chunk_bytes", [(512, 640), (1022, 640), (1023, 768), (1024, 768)])
def test_sliced_ell_byte_chunks_and_12_bit_column_words(pkg, oracle, monkeypatch, cols, chunk_bytes):
    """Byte chunks (TKSPMV_Q1_7_F32: Q1.7 values rounded to nearest): with at most 1022 columns the column words take 12 bits
    (640-byte chunks, padding slots at columns 1022 / 1023), else 16 (768 bytes); TKSPMV_SELL_C12=0 keeps 16 bits. Either way
    decode(pack(A)) returns every entry with its rounded value."""
    m = pkg.generate_matrix(6000, cols, 30, "gamma", 8)
    want = oracle.round_to_q17(m.val)
    for env, nbytes in ((None, chunk_bytes), ("0", 768)):
        if env is not None:
            monkeypatch.setenv("TKSPMV_SELL_C12", env)
        r, c, v, info = pkg.sell_roundtrip(m, 64, precision=pkg.Q1_7_F32)
        monkeypatch.delenv("TKSPMV_SELL_C12", raising=False)
        assert info["stream_bytes"] == info["chunks"] * nbytes
        order = np.argsort(r, kind="stable")
        assert np.array_equal(r[order], m.row) and np.array_equal(c[order], m.col)
        assert np.array_equal(v[order].view(np.uint32), want.view(np.uint32))


def test_sliced_ell_edge_cases(pkg):
    """Empty rows vanish (they can never be candidates), a single row, a row longer than any chunk, one column."""
    rng = np.random.RandomState(1)
    lens = [0, 5, 0, 0, 1, 700, 2, 0]
    row = np.repeat(np.arange(len(lens)), lens).astype(np.uint32)
    col = rng.randint(0, 40, row.shape[0]).astype(np.uint32)
    val = rng.rand(row.shape[0]).astype(np.float32)
    m = pkg.CooMatrix(len(lens), 40, row, col, val)
    r, c, v, info = pkg.sell_roundtrip(m, 16)
    assert info["slices"] == 1 and info["chunks"] == 16 and info["partitions"] == 1  # 700 entries = 11 lanes of 64, + 3 rows
    assert r.tolist()[:700] == [5] * 700  # the longest rows lead
    order = np.argsort(r, kind="stable")
    assert np.array_equal(r[order], row) and np.array_equal(c[order], col) and np.array_equal(v[order], val)
    empty = pkg.CooMatrix(10, 8, np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32))
    r, c, v, info = pkg.sell_roundtrip(empty, 4)
    assert r.shape[0] == 0 and info["chunks"] == 0
    wide = pkg.generate_matrix(100, 2000, 5, "uniform", 1)
    with pytest.raises(pkg.TkspmvError):
        pkg.sell_roundtrip(wide, 4)


def test_packed_file_rejects_damage(pkg, tmp_path):
    m = pkg.generate_matrix(500, 64, 8, "uniform", 3)
    p = pkg.Packed(m, n_wave_partitions=8)
    path = tmp_path / "m.tkspmv"
    p.save(path)
    raw = bytearray(path.read_bytes())
    cases = {
        "missing": None,
        "truncated": bytes(raw[: len(raw) - 100]),
        "trailing": bytes(raw) + b"\0\0\0\0",
        "bad magic": b"XXXXXXXX" + bytes(raw[8:]),
        "flipped payload bit": bytes(raw[:1000]) + bytes([raw[1000] ^ 0x10]) + bytes(raw[1001:]),
        "header only": bytes(raw[:128]),
        "empty": b"",
    }
    for name, blob in cases.items():
        f = tmp_path / (name.replace(" ", "_") + ".tkspmv")
        if blob is not None:
            f.write_bytes(blob)
        with pytest.raises(pkg.TkspmvError) as ei:
            pkg.Packed.load(f)
        assert ei.value.status == pkg._lib.ERR_IO, name
    # an inconsistent header (packet count raised) must not lead to out-of-bounds reads either
    bad = bytearray(raw)
    bad[28:32] = (int.from_bytes(raw[28:32], "little") + 7).to_bytes(4, "little")  # n_packets
    f = tmp_path / "header.tkspmv"
    f.write_bytes(bytes(bad))
    with pytest.raises(pkg.TkspmvError):
        pkg.Packed.load(f)


def test_half_conversion_is_ieee_round_to_nearest_even(pkg, oracle):
    """TKSPMV_F16's value stream: the packer's and the oracle's float -> half conversion against numpy.float16 (IEEE
    binary16, round to nearest even, overflow to infinity), including subnormals, ties and the overflow boundary."""
    rng = np.random.RandomState(3)
    special = np.array([0.0, 1.0, 0.5, 1.0 / 3.0, 65504.0, 65519.9, 65520.0, 1e5, 6.1035156e-05, 6.0e-05, 5.9604645e-08,
                        2.9802322e-08, 2.9802326e-08, 1e-9, 1.0009765625, 1.00048828125, 1.00146484375, 0.99951171875,
                        2049.0, 2051.0, 4097.0], dtype=np.float32)
    vals = np.concatenate([special, rng.rand(5000).astype(np.float32), (rng.rand(2000) * 1e-4).astype(np.float32),
                           (rng.rand(2000) * 7e4).astype(np.float32), (10.0 ** rng.uniform(-9, 5, 3000)).astype(np.float32)])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).astype(np.float32)
    got = oracle.round_to_half(vals)
    finite = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), finite)
    assert np.array_equal(got[finite].view(np.uint32), want[finite].view(np.uint32))
    # the packer writes the same halves: one row per value, decode gives them back
    n = 3000
    m = pkg.CooMatrix(n, 8, np.arange(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32), vals[:n].copy())
    p = pkg.Packed(m, n_wave_partitions=16, precision=pkg.F16)
    r, c, v = p.decode()
    assert np.array_equal(r, m.row) and np.array_equal(v.view(np.uint32), want[:n].view(np.uint32))
    assert p.info()["packed_bytes"] < pkg.Packed(m, n_wave_partitions=16).info()["packed_bytes"]
