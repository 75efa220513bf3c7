"""Round-4 diagnostics of ONE query per launch (run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations):
the fused single launch (stream_kernel), one query through the batch kernel, and a one-pass load-only launch of the same
geometry -- what a single launch costs before any arithmetic. Development probe."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

mod = _pkg.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
what = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fused", "batch1", "read1"]
m = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(16)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=4)
out_i = torch.zeros(16, 100, dtype=torch.int32, device="cuda")
out_v = torch.zeros(16, 100, dtype=torch.float32, device="cuda")
s = torch.cuda.Stream()


def timed(fn, count):
    with torch.cuda.stream(s):
        for i in range(10):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for i in range(count):
            fn(i)
        e1.record(s)
    s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / count


if "fused" in what:
    t = timed(lambda i: eng.enqueue(dxs[i % 16].data_ptr(), out_i[i % 16].data_ptr(), out_v[i % 16].data_ptr(), s.cuda_stream), n)
    print(f"{rows} rows, fused single launch back to back: {t:.2f} us per query")
    own = []
    for i in range(n):
        eng.reset_device(dxs[i % 16].data_ptr())
        own.append(eng() / 1e3)
    print(f"{rows} rows, tkspmv_run (device-side stamp, or events): median {np.median(own[2:]):.2f} us, p95 {np.percentile(own[2:], 95):.2f}")
if "batch1" in what:
    t = timed(lambda i: eng.enqueue_batch(dxs[i % 16].data_ptr(), 1, out_i[i % 16].data_ptr(), out_v[i % 16].data_ptr(), s.cuda_stream), n)
    print(f"{rows} rows, batch kernel with ONE query per launch back to back: {t:.2f} us per query; counters {eng.debug_counters()}")
if "read1" in what:
    r = [eng.time_stream_read(1) / 1e3 for _ in range(max(n // 4, 5))]
    print(f"{rows} rows, load-only launch of one pass (event bracket): median {np.median(r):.2f} us, min {min(r):.2f}")
    print(f"{rows} rows, load-only 64 passes: {eng.time_stream_read(64) / 1e3:.2f} us per pass")
eng.close()
