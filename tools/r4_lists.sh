#!/bin/bash
set -u
cat > /tmp/exact.py <<'PY'
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import _pkg
mod = _pkg.load()
rows = int(sys.argv[1])
m = mod.generate_matrix(rows, 1024, 20, "gamma", 2)
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 1000 + i) for i in range(64)])
dxs = torch.from_numpy(xs).cuda()
eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=4)
eng.enqueue_many(dxs.data_ptr(), 64, 1024); eng.synchronize()
t = sorted(eng.time_queries(dxs.data_ptr(), 64, 512) / 1e3 for _ in range(7))
i = eng.info()
print(f"{os.path.basename(os.getcwd()):8s} rows {rows} LOCAL={os.environ.get('TKSPMV_LOCAL','-')} SELECTORS={os.environ.get('TKSPMV_SELECTORS','-')} SMALL={os.environ.get('TKSPMV_SMALL_PACKETS','-')}: median {t[3]:.2f} us/query (min {t[0]:.2f})  mode {i['batch_mode'] & 0xFFFF:#x} parts {i['n_wave_partitions']}", flush=True)
eng.close()
PY
for r in 1 2; do for l in 4 2 1; do TKSPMV_LOCAL=0 TKSPMV_OVF_LISTS=$l timeout -k 10 120 python3 /tmp/exact.py 1000000 2>&1 | grep rows | sed "s/$/ lists $l/" || exit 1; done; done
timeout -k 10 120 python3 /tmp/exact.py 1000000 2>&1 | grep rows
timeout -k 10 200 python3 bench.py --steps 640 --warmup 64 --cpu-seconds 0 --skip-warm --traffic off > gpurun_out/lists_bench.json 2> gpurun_out/lists_bench.err || { tail -5 gpurun_out/lists_bench.err; exit 1; }
python3 - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/lists_bench.json") if l.startswith("{")][-1])
print("kernel_us", j["roofline"].get("kernel_us"), "frac", j["roofline"]["frac"], "state", j.get("exchange_state_bytes"))
print("nonstationary", json.dumps(j.get("nonstationary"))[:400])
PY
bash tools/r4_gputests.sh gputests_b
