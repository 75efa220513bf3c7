"""The cases of tests/test_gpu_local_fuzz.py for seeds beyond the committed 32 (a one-off soak after kernel changes): the default
engine against the engine of the device-wide exchange on the same partition cut, bit for bit, two passes per case.
  [BIG=1] python tools/fuzz_soak.py FIRST LAST"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

pkg = _pkg.load()
first, last = int(sys.argv[1]), int(sys.argv[2])
bad_cases = 0
for seed in range(first, last):
    rng = np.random.default_rng(1000 + seed)
    rows = int(rng.choice([900, 5000, 23000, 70000, 160000, 330000]))
    if os.environ.get("BIG"):  # (round 5: the checked local thresholds serve every size -- BIG=1 draws 1.5M .. 4M rows of fp32 / fp16 values)
        rows = int(rng.choice([1500000, 2500000, 4000000]))
    cols = int(rng.choice([64, 300, 512, 1024]))
    nnz = int(rng.choice([1, 3, 12, 20, 45]))
    k = int(rng.choice([1, 8, 100, 100, 250]))
    prec = str(rng.choice(["F32", "F32", "F16", "Q1_7", "Q1_7_WIDE", "FIXED"]))
    dist = str(rng.choice(["gamma", "uniform"]))
    if os.environ.get("BIG"):
        prec = str(rng.choice(["F32", "F32", "F16"]))
        nnz = int(rng.choice([8, 20]))
    elif rows * nnz > 8_000_000:
        nnz = max(1, 8_000_000 // rows)
    m = pkg.generate_matrix(rows, cols, nnz, dist, 50 + seed)
    nq = 40
    xs = np.stack([pkg.create_sample_vector(cols, True, False, True, 3000 + 97 * seed + i) for i in range(nq)])
    if prec in ("Q1_7", "Q1_7_WIDE"):
        xs = (xs * np.float32(30.0)).astype(np.float32)
    scale = rng.choice([1.0, 1.0, 0.2, 3.0], size=nq).astype(np.float32)
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
    for mode in ("exchange", "default"):
        pkg.set_option("LOCAL", "0" if mode == "exchange" else None)
        pkg.set_option("PACE", "0" if mode == "exchange" else None)
        eng = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, **kw)
        oi = torch.full((nq, k), -1, dtype=torch.int32, device="cuda")
        ov = torch.full((nq, k), -1.0, dtype=torch.float32, device="cuda")
        for rep in range(2):
            eng.enqueue_batch(dxs.data_ptr(), nq, oi.data_ptr(), ov.data_ptr())
            eng.synchronize()
        res[mode] = (oi.cpu().numpy().copy(), ov.cpu().numpy().copy(), eng.info()["batch_mode"], eng.debug_counters()["checks_failed"])
        eng.close()
    pkg.set_option("LOCAL", None)
    pkg.set_option("PACE", None)
    ex, df = res["exchange"], res["default"]
    bad = [q for q in range(nq) if not (np.array_equal(ex[0][q], df[0][q]) and np.array_equal(ex[1][q].view(np.uint32), df[1][q].view(np.uint32)))]
    if bad:
        bad_cases += 1
    print(f"seed {seed}: rows={rows} cols={cols} nnz={nnz} k={k} {prec} {dist} mode {df[2] & 0xFFFF:#x} failed checks {df[3]}: {'DIFFER ' + str(bad) if bad else 'equal'}", flush=True)
print("cases that differ:", bad_cases)
sys.exit(1 if bad_cases else 0)
