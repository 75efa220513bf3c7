"""bench.py's output contract, as the round driver consumes it: ONE JSON line with the named keys, the headline config, a
roofline object computed from the timed region's own events and a parity verdict -- on the driver's own command line
(`--gpus 1 --steps 20 --warmup 5`) and on the row-sharded workload of `--gpus N` rehearsed with one rank. Also the
load-only probe behind `roofline.read_only` (tkspmv_time_stream_read)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_headline_line_on_the_drivers_command_line():
    d = _bench("--gpus", "1", "--steps", "20", "--warmup", "5", "--skip-warm", "--cpu-seconds", "0", "--traffic", "off")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "parity_checked", "timing"):
        assert key in d, key
    assert d["metric"] == "queries_per_sec" and d["unit"] == "queries/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["config"]["rows"] == 1000000 and d["config"]["cols"] == 1024 and d["config"]["k"] == 100 and "workload" in d["config"]
    assert d["parity_checked"] is True and d["parity"]["bit_exact_vs_order_matched_oracle"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["kernel_us"] * 1e3)) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.5 < r["frac"] < 1.0
    # the host clock contains the device time of the same region
    assert d["ms_per_step"] * 1e3 >= r["kernel_us"] and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    ro = r["read_only"]
    # (the unpaced probe is no floor on every box -- its XCDs are served unevenly --: the paced product may come within a few per cent
    #  of it or beat it; the probe on a timetable, `paced`, is one)
    assert 0.6 < ro["frac_of_peak"] < 1.0 and 0.7 < ro["headline_kernel_vs_read_only"] <= 1.05
    assert ro["paced"]["us_per_pass"] <= ro["us_per_pass"] * 1.001 and 0.6 < ro["paced"]["frac_of_peak"] < 1.0
    assert 0.7 < r["kernel_vs_read_only_paced"] < 1.0 and abs(r["read_only_paced_us"] - ro["paced"]["us_per_pass"]) < 1e-9


def test_sharded_line_rehearsed_with_one_rank():
    d = _bench("--total-rows", "600000", "--steps", "40", "--warmup", "8", env={"TKSPMV_BENCH_CROSS": "1"})
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["config"]["rows"] == 600000 and d["config"]["n_shards"] == 1
    assert d["parity_checked"] is True and d["exchange"]["cross_check"]["native_vs_torch_exchange_same_list"] is True
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["rows"] == 600000 and d["roofline"]["bound"] == "hbm"


def test_load_only_probe_brackets_the_kernel(pkg):
    import torch
    m = pkg.generate_matrix(400000, 1024, 20, "gamma", 6)
    xs = np.stack([pkg.create_sample_vector(1024, True, False, True, 70 + i) for i in range(8)])
    dxs = torch.from_numpy(xs).cuda()
    eng = pkg.SpMV(m.row, m.col, m.val, m.rows, m.cols, vec=xs[0], k=100, device=0, stream_replicas=8)  # 8 x 47 MB: no cache holds them
    info = eng.info()
    stream_bytes = info["n_packets"] * 1408  # fp32 values + 12-bit column words
    eng.time_queries(dxs.data_ptr(), 8, 64)
    t_kernel = min(eng.time_queries(dxs.data_ptr(), 8, 256) for _ in range(3))
    t_read = sorted(eng.time_stream_read(32) for _ in range(5))[2]
    # not faster than twice the 8 TB/s specification (part of a 47 MB stream copy may still sit in the Infinity Cache), and
    # below the kernel that also computes
    assert 0.5 * stream_bytes / 8000.0 < t_read < t_kernel
    with pytest.raises(pkg.TkspmvError):
        eng.time_stream_read(0)
    eng.close()
