"""Load-only floor of the packet stream with static partitions (every wave owns one) against dynamic claiming (every wave
takes its next partition from a counter), over partition sizes: what is dynamic balancing across the XCDs worth?
  python tools/claim_probe.py [partitions-per-wave ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

torch.cuda.init()
mod = _pkg.load()
m = mod.generate_matrix(1000000, 1024, 20, "gamma", 2)
x = mod.create_sample_vector(1024, True, False, True, 1)
for mult in [int(v) for v in (sys.argv[1:] or ["1", "2", "4"])]:
    n_parts = mod.Packed.wave_partitions(0) * mult
    os.environ["TKSPMV_PARTITIONS_HINT"] = str(n_parts)
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=x, k=100, device=0, stream_replicas=4)
    for mp in ("0", "4", "0", "4"):
        os.environ["TKSPMV_READ_PROBE_MAP"] = mp
        r = sorted(eng.time_stream_read(64) / 1e3 for _ in range(5))[2]
        print(f"partitions per wave {mult} ({eng.info()['n_wave_partitions']} partitions): map {mp}: {r:6.2f} us per pass")
    eng.close()
