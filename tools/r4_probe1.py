"""Round-4 probe: what ONE pass costs as a launch of its own -- the load-only kernel of the engine's geometry at several prefetch
depths, with and without the engine's per-packet arithmetic (TKSPMV_READ_PROBE=depth,work; 1536-byte packets). Run under
rocprofv3 --kernel-trace --stats: every variant is its own template instantiation. Development probe."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TKSPMV_F32_C12"] = "0"
import _pkg  # noqa: E402

mod = _pkg.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0, stream_replicas=4)
for v in ("2,0", "3,0", "4,0", "6,0", "8,0", "16,0", "3,-1", "3,-2", "4,-2"):
    os.environ["TKSPMV_READ_PROBE"] = v
    r = [eng.time_stream_read(1) / 1e3 for _ in range(20)]
    r64 = eng.time_stream_read(64) / 1e3
    print(f"depth,work = {v:6s}: one-pass launch (event bracket) median {np.median(r):6.2f} us, min {min(r):6.2f}; 64 passes: {r64:6.2f} us per pass", flush=True)
eng.close()
