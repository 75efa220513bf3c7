"""fp32 values with 12-BIT column words (csrc/wbscsr.hpp F32C12; the default up to 1024 columns, TKSPMV_F32_C12=0 keeps 16 bits): 8.3 % fewer bytes in the
stream, the same fp32 arithmetic in the same order. Every path that streams the packets must return the same bits as with
16-bit column words; the device packer must produce the host packer's bytes; a packed file must round-trip."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engines(pkg, monkeypatch, m, x, **kw):
    monkeypatch.setenv("TKSPMV_F32_C12", "0")
    e16 = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=x, device=0, **kw)
    monkeypatch.delenv("TKSPMV_F32_C12")
    e12 = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=x, device=0, **kw)
    return e16, e12


@pytest.mark.parametrize("rows,cols,nnz,k,dist", [(200000, 1024, 20, 100, "gamma"), (30000, 512, 40, 8, "uniform"), (1000000, 1024, 20, 100, "gamma"),
                                                  (700, 64, 5, 3, "uniform")])
def test_same_bits_as_16_bit_column_words(pkg, oracle, monkeypatch, rows, cols, nnz, k, dist):
    import torch
    m = pkg.generate_matrix(rows, cols, nnz, dist, 21)
    xs = np.stack([pkg.create_sample_vector(cols, True, False, True, 300 + i) for i in range(40)])
    e16, e12 = _engines(pkg, monkeypatch, m, xs[0], k=k, stream_replicas=2)
    i16, i12 = e16.info(), e12.info()
    assert i12["precision"] == i16["precision"] == pkg.F32 and i12["algorithmic_bytes"] == i16["algorithmic_bytes"]
    assert i12["n_packets"] == i16["n_packets"] and 0.91 < i12["packed_bytes"] / i16["packed_bytes"] < 0.93
    # one query at a time (fused single launches)
    for q in range(3):
        out = []
        for e in (e16, e12):
            e.reset(xs[q])
            e()
            out.append(e.read_result())
        assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    gi, gv = oracle.gold_topk(m.row, m.col, m.val, xs[2], k)
    npos = int((gv > 0).sum())
    assert set(out[1][1][:npos].tolist()) == set(gi[:npos].tolist())
    # a batch of 40 queries back to back
    dxs = torch.from_numpy(xs).cuda()
    res = []
    for e in (e16, e12):
        oi = torch.full((40, k), -1, dtype=torch.int32, device="cuda")
        ov = torch.full((40, k), -1.0, dtype=torch.float32, device="cuda")
        e.enqueue_batch(dxs.data_ptr(), 40, oi.data_ptr(), ov.data_ptr())
        e.synchronize()
        res.append((oi.cpu().numpy(), ov.cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
    # the full score vector (SpMV-only kernel)
    e16.reset(xs[5])
    e12.reset(xs[5])
    assert np.array_equal(e16.scores().view(np.uint32), e12.scores().view(np.uint32))
    e16.close()
    e12.close()


def test_device_packer_and_file_round_trip(pkg, monkeypatch, tmp_path):
    m = pkg.generate_matrix(150000, 1000, 25, "gamma", 3)
    host = pkg.Packed(m, k=100, nnz_per_lane=4, n_wave_partitions=512)
    dev = pkg.Packed(m, k=100, nnz_per_lane=4, n_wave_partitions=512, on_device=True)
    rh, rd = host.raw(), dev.raw()
    assert rh[1] == rd[1] == 1408
    for a, b in zip((rh[0], rh[2], rh[3], rh[4]), (rd[0], rd[2], rd[3], rd[4])):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    r, c, v = dev.decode()
    assert np.array_equal(r, m.row) and np.array_equal(c, m.col) and np.array_equal(v, m.val)
    path = str(tmp_path / "m.tkspmv")
    host.save(path)
    x = pkg.create_sample_vector(1000, True, False, True, 8)
    back = pkg.Packed.load(path)
    e_file = pkg.SpMV.from_packed(back, k=100, vec=x, device=0)
    e_mem = pkg.SpMV.from_packed(host, k=100, vec=x, device=0)  # (same packing: same summation order, same bits)
    assert e_file.info()["packed_bytes"] < 0.93 * 6.3 * m.nnz
    out = []
    for e in (e_file, e_mem):
        e()
        out.append(e.read_result())
        e.close()
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
