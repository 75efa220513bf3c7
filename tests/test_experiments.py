"""Accuracy metrics and result-file handling of the experiment driver -- no GPU needed."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ex(pkg):
    from importlib import import_module
    return import_module("approximate_spmv_topk_amd.experiments")


def test_metrics_match_the_reference_functions(ex):
    """Golden vectors produced by executing plot_errors.py's kendall_tau / ndcg (tests/golden/make_golden_metrics.py)."""
    gold = json.load(open(os.path.join(HERE, "golden", "metrics_golden.json")))
    assert len(gold["cases"]) >= 8
    for c in gold["cases"]:
        assert ex.kendall_tau(c["sw_idx"], c["hw_idx"]) == pytest.approx(c["kendall"], rel=1e-12, abs=1e-12)
        got = ex.ndcg(c["sw_idx"], c["sw_val"], c["hw_idx"], c["hw_val"])
        assert np.allclose(got, c["ndcg"], rtol=1e-12, atol=0)


def test_metric_properties(ex):
    a = list(range(20))
    v = [1.0 - 0.01 * i for i in range(20)]
    assert ex.precision_at(a, a, 10) == 1.0 and ex.kendall_tau(a, a) == 1.0 and ex.ndcg(a, v, a, v)[0] == pytest.approx(1.0)
    assert ex.kendall_tau(a, a[::-1]) == -1.0
    assert ex.precision_at(a, list(range(100, 120)), 20) == 0.0 and ex.ndcg(a, v, list(range(100, 120)), v)[0] == 0.0
    assert ex.precision_at(a, a[10:] + a[:10], 10) == 0.0 and ex.precision_at(a, a[10:] + a[:10], 20) == 1.0


def test_result_csv_round_trip(ex, tmp_path):
    k = 4
    lines = [",".join(ex.GPU_COLUMNS)]
    for it in range(5):
        sw_i, hw_i = [9, 7, 5, 3], [9, 5, 7, 3] if it % 2 else [9, 7, 5, 3]
        lines.append(",".join(str(x) for x in (it, 0, 0, 1.5, 0.5, 10.0, 0.02, 0.03 + 0.001 * it, 0.01, k,
                                               ";".join(map(str, sw_i)), "0.9;0.8;0.7;0.6", ";".join(map(str, hw_i)),
                                               "0.9;0.8;0.7;0.6")))
    p = tmp_path / "mi355x_10_8_gamma_2_f32_4_5.csv"
    p.write_text("\n".join(lines) + "\n")
    rows = ex.read_result_csv(str(p))
    assert len(rows) == 5 and rows[1]["hw_res_idx"] == [9, 5, 7, 3] and rows[0]["k"] == 4
    acc = ex.accuracy(rows, thresholds=(1, 4), skip=2)
    assert acc["iterations"] == 3 and acc["prec_4"] == 1.0 and acc["prec_1"] == 1.0
    assert acc["kendall_4"] == pytest.approx((1.0 + (4 / 6) + 1.0) / 3)  # iteration 3 swaps one pair: tau = (5 - 1) / 6
    assert acc["hw_exec_time_ms_mean"] == pytest.approx(0.033)
    with pytest.raises(ValueError):
        (tmp_path / "bad.csv").write_text("a,b\n1,2\n")
        ex.read_result_csv(str(tmp_path / "bad.csv"))
    assert ex.matrix_name(10000, 1024, 20, "gamma") == "matrix_10000_1024_20_gamma.mtx"
