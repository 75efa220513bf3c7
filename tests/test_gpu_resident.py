"""desc.impl = TKSPMV_IMPL_RESIDENT: the reference hosts' loop -- reset(vec), operator()(), read_result(), one query at a
time (src/fpga/src/host_spmv_bscsr.cpp:602-632) -- served by ONE resident launch through pinned memory (no kernel launch,
no copy engine, no stream synchronisation per query). Results must be bit-identical to the ordinary engine's; every other
entry point makes the resident kernel leave first; the kernel leaves by itself when idle and is started again on demand."""
import os
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_resident_loop_is_bit_identical_and_survives_everything_else(pkg, oracle, monkeypatch):
    import torch
    m = pkg.generate_matrix(300000, 1024, 20, "gamma", 21)
    xs = np.stack([pkg.create_sample_vector(1024, True, False, True, 60 + i) for i in range(12)])
    ref = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=2)
    want = []
    for x in xs:
        ref.reset(x)
        ref()
        want.append(ref.read_result())
    ref.close()
    monkeypatch.setenv("TKSPMV_RESIDENT_IDLE_MS", "30")
    eng = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=2, impl=pkg._lib.IMPL_RESIDENT)
    kern = []
    for rep in range(3):
        for q, x in enumerate(xs):
            eng.reset(x)
            kern.append(eng())
            val, idx = eng.read_result()
            assert np.array_equal(idx, want[q][1]), (rep, q)
            assert np.array_equal(val.view(np.uint32), want[q][0].view(np.uint32)), (rep, q)
    assert 5e3 < np.median(kern) < 2e5  # device time of a query, as reported by the kernel (ns)
    gi, gv = oracle.gold_topk(m.row, m.col, m.val, xs[-1], 100)
    assert set(idx.tolist()) == set(gi.tolist()) and np.allclose(val, gv, rtol=1e-4, atol=0)
    # the idle timeout: the kernel leaves, the next query starts it again
    time.sleep(0.15)
    eng.reset(xs[3])
    eng()
    val, idx = eng.read_result()
    assert np.array_equal(idx, want[3][1]) and np.array_equal(val.view(np.uint32), want[3][0].view(np.uint32))
    # any other entry point: the resident kernel leaves first, and comes back for the next tkspmv_run
    dxs = torch.from_numpy(xs).cuda()
    out_i = torch.zeros(12, 100, dtype=torch.int32, device="cuda")
    out_v = torch.zeros(12, 100, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.enqueue_batch(dxs.data_ptr(), 12, out_i.data_ptr(), out_v.data_ptr())
    eng.synchronize()
    for q in range(12):
        assert np.array_equal(out_i[q].cpu().numpy().view(np.uint32), want[q][1])
    y = eng.scores()  # (uses the x of the last set_query, which a resident engine had left in pinned memory)
    assert abs(float(y[want[3][1][0]]) - float(want[3][0][0])) <= 1e-6 * float(want[3][0][0])
    eng.reset(xs[5])
    eng()
    val, idx = eng.read_result()
    assert np.array_equal(idx, want[5][1]) and np.array_equal(val.view(np.uint32), want[5][0].view(np.uint32))
    eng.reset_device(dxs[7].data_ptr())  # x already on the device: the ordinary single launch
    eng()
    val, idx = eng.read_result()
    assert np.array_equal(idx, want[7][1])
    eng.reset(xs[8])
    eng()
    val, idx = eng.read_result()
    assert np.array_equal(idx, want[8][1])
    eng.close()  # (while the kernel is resident)


def test_engines_the_resident_kernel_does_not_apply_to_run_the_default(pkg):
    m = pkg.generate_matrix(20000, 3000, 20, "gamma", 5)  # more than 1024 columns
    x = pkg.create_sample_vector(3000, True, False, True, 3)
    a = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=x, k=50, device=0, impl=pkg._lib.IMPL_RESIDENT)
    b = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=x, k=50, device=0)
    a()
    b()
    assert np.array_equal(a.read_result()[1], b.read_result()[1])
    a.close()
    b.close()


def test_drop_in_executable_with_the_resident_kernel(pkg, tmp_path):
    exe = os.path.join(ROOT, "bin", "approximate-spmv-mi355x-topk")
    g = pkg.generate_matrix(100000, 1024, 20, "gamma", 1)
    p = tmp_path / "matrix_100000_1024_20_gamma.mtx"
    pkg.write_mtx(str(p), g, index_base=0)
    env = dict(os.environ, TKSPMV_SEED="5")
    r = subprocess.run([exe, "-t", "8", "-m", str(p), "-k", "100", "-i", "3", "-r"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    assert len(lines) == 9
    for ln in lines[1:]:
        f = ln.split(",")
        assert set(f[10].split(";")) == set(f[12].split(";")) and int(f[2]) == 0
