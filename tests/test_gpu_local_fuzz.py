"""Round-3 modes of the batch kernel on seeded random shapes (sizes, k, value types, query scales drawn per case). Three engines per
case: round 3's defaults; the same geometry with the device-wide exchange (TKSPMV_LOCAL=0, TKSPMV_PACE=0: same partition cut, so
every query must be BIT-identical -- rows, scores, order); round 2's behaviour (TKSPMV_SMALL_PACKETS=0: one selector, partitions of
4+ packets -- another cut, so a row's fp32 sum may differ in its last bits: same rows up to ties at that precision, scores within
1e-5). The gold comparison of these engines lives in the other GPU tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", list(range(32)))
def test_round3_batch_modes_equal_the_device_wide_exchange(pkg, monkeypatch, seed):
    import torch
    rng = np.random.default_rng(1000 + seed)
    rows = int(rng.choice([900, 5000, 23000, 70000, 160000, 330000]))
    cols = int(rng.choice([64, 300, 512, 1024]))
    nnz = int(rng.choice([1, 3, 12, 20, 45]))
    k = int(rng.choice([1, 8, 100, 100, 250]))
    prec = str(rng.choice(["F32", "F32", "F16", "Q1_7", "Q1_7_WIDE", "FIXED"]))
    dist = str(rng.choice(["gamma", "uniform"]))
    if rows * nnz > 8_000_000:
        nnz = max(1, 8_000_000 // rows)
    m = pkg.generate_matrix(rows, cols, nnz, dist, 50 + seed)
    nq = 40
    xs = np.stack([pkg.create_sample_vector(cols, True, False, True, 3000 + 97 * seed + i) for i in range(nq)])
    if prec in ("Q1_7", "Q1_7_WIDE"):
        xs = (xs * np.float32(30.0)).astype(np.float32)
    scale = rng.choice([1.0, 1.0, 0.2, 3.0], size=nq).astype(np.float32)  # queries change scale: carried thresholds must cope
    xs = xs * scale[:, None]
    if seed % 3 == 0:
        xs[rng.integers(0, nq)] = 0.0
    if seed % 4 == 1:
        xs[rng.integers(0, nq)] *= np.float32(-1.0)
    dxs = torch.from_numpy(np.ascontiguousarray(xs)).cuda()
    kw = dict(k=k, device=0, precision=getattr(pkg, prec))
    if prec == "FIXED":
        kw["fixed_width"] = int(rng.choice([12, 20, 25, 32]))
    res = {}
    for mode in ("round2", "exchange", "round3"):
        for var in ("TKSPMV_SMALL_PACKETS", "TKSPMV_LOCAL", "TKSPMV_PACE"):
            monkeypatch.delenv(var, raising=False)
        if mode == "round2":
            monkeypatch.setenv("TKSPMV_SMALL_PACKETS", "0")
        elif mode == "exchange":
            monkeypatch.setenv("TKSPMV_LOCAL", "0")
            monkeypatch.setenv("TKSPMV_PACE", "0")
        eng = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, **kw)
        oi = torch.full((nq, k), -1, dtype=torch.int32, device="cuda")
        ov = torch.full((nq, k), -1.0, dtype=torch.float32, device="cuda")
        for rep in range(2):  # (the second pass starts from the thresholds the first one left behind)
            eng.enqueue_batch(dxs.data_ptr(), nq, oi.data_ptr(), ov.data_ptr())
            eng.synchronize()
        res[mode] = (oi.cpu().numpy().view(np.uint32), ov.cpu().numpy(), eng.info()["batch_mode"], eng.debug_counters())
        eng.close()
    r2, ex, r3 = res["round2"], res["exchange"], res["round3"]
    desc = f"rows={rows} cols={cols} nnz/row={nnz} k={k} {prec} {dist}: batch_mode {r3[2] & 0xFFFF:#x}, {r3[3]}"
    bad = [q for q in range(nq) if not (np.array_equal(ex[0][q], r3[0][q]) and np.array_equal(ex[1][q].view(np.uint32), r3[1][q].view(np.uint32)))]
    assert not bad, (desc, bad)
    for q in range(nq):  # another partition cut: scores to 1e-5, the same rows wherever scores are not tied at that precision
        assert np.allclose(r2[1][q], r3[1][q], rtol=1e-5, atol=1e-30), (desc, q)
        diff = set(r2[0][q].tolist()) ^ set(r3[0][q].tolist())
        if diff:
            kth = float(r3[1][q][-1])
            sc = {int(i): float(v) for i, v in zip(r2[0][q], r2[1][q])}
            sc.update({int(i): float(v) for i, v in zip(r3[0][q], r3[1][q])})
            assert all(abs(sc[r] - kth) <= 2e-5 * max(abs(kth), 1e-30) for r in diff), (desc, q, sorted(diff)[:6])


@pytest.mark.parametrize("lists", [1, 2, 4])
def test_overflow_lists_shared_by_the_queries_of_a_launch(pkg, monkeypatch, lists):
    """The exact kernel's overflow lists are shared round robin by the queries of a launch under flow control (OVF_LISTS: 2 behind
    the kernel of local thresholds, 4 else). Round 4 found two selections on ONE list at a time mixing their candidates when the
    lists were fewer than the selector workgroups (330 000 x 512 Q1.7, k = 1: a query reported another query's best row). The case
    that showed it -- half of the checks fail, the gate closes, every query of the second pass goes through the exact kernel --
    with 1, 2 and 4 lists against the engine of the device-wide exchange, bit for bit, twice."""
    import torch
    m = pkg.generate_matrix(330000, 512, 24, "gamma", 51)
    nq, k = 40, 1
    rng = np.random.default_rng(1001)
    xs = np.stack([pkg.create_sample_vector(512, True, False, True, 3097 + i) for i in range(nq)]) * np.float32(30.0)
    xs = (xs * rng.choice([1.0, 1.0, 0.2, 3.0], size=nq).astype(np.float32)[:, None]).astype(np.float32)
    xs[7] *= np.float32(-1.0)
    dxs = torch.from_numpy(np.ascontiguousarray(xs)).cuda()

    def run():
        eng = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=k, device=0, precision=pkg.Q1_7)
        oi = torch.full((nq, k), -1, dtype=torch.int32, device="cuda")
        ov = torch.full((nq, k), -1.0, dtype=torch.float32, device="cuda")
        out = []
        for rep in range(3):
            eng.enqueue_batch(dxs.data_ptr(), nq, oi.data_ptr(), ov.data_ptr())
            eng.synchronize()
            out.append((oi.cpu().numpy().copy(), ov.cpu().numpy().copy()))
        c = eng.debug_counters()
        eng.close()
        return out, c

    monkeypatch.setenv("TKSPMV_LOCAL", "0")
    ref, _ = run()
    monkeypatch.delenv("TKSPMV_LOCAL")
    monkeypatch.setenv("TKSPMV_OVF_LISTS", str(lists))
    out, c = run()
    for rep in range(3):
        assert np.array_equal(ref[0][0], out[rep][0]) and np.array_equal(ref[0][1].view(np.uint32), out[rep][1].view(np.uint32)), (lists, rep, c)
    assert c["checks_failed"] > 0, c  # (the exact kernel did run behind the local one)
