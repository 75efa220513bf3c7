#!/usr/bin/env python3
"""Golden vectors for the accuracy metrics (experiments.py) from the REFERENCE's own functions.

plot_errors.py cannot be imported here (seaborn / matplotlib styles are missing), so the two function definitions are
taken out of its syntax tree and executed on their own, with numpy as their only global -- reference code is executed in
this container only, nothing of it is stored; what is committed are inputs and outputs (metrics_golden.json).
Run from the repository root:  python tests/golden/make_golden_metrics.py
"""
import ast
import json
import os

import numpy as np

REF = "/root/reference/src/resources/python/plotting/plot_errors.py"
tree = ast.parse(open(REF).read())
wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("kendall_tau", "ndcg")]
ns = {"np": np}
exec(compile(ast.Module(body=wanted, type_ignores=[]), REF, "exec"), ns)

rng = np.random.RandomState(11)
cases = []
for n, overlap in ((8, 8), (8, 5), (16, 16), (16, 3), (50, 44), (100, 100), (100, 71), (5, 0)):
    universe = rng.permutation(10 * n + 10)
    sw = universe[:n].tolist()
    hw = sw[:overlap] + universe[n:2 * n - overlap].tolist()
    perm = rng.permutation(n)
    if overlap == n and n != 16:
        perm = np.arange(n)  # identical lists
    hw = [hw[i] for i in perm]
    sw_val = np.sort(rng.rand(n))[::-1].round(6).tolist()
    hw_val = np.sort(rng.rand(n))[::-1].round(6).tolist()
    nd = ns["ndcg"](sw, sw_val, hw, hw_val)
    cases.append({"sw_idx": [int(x) for x in sw], "sw_val": sw_val, "hw_idx": [int(x) for x in hw], "hw_val": hw_val,
                  "kendall": float(ns["kendall_tau"](sw, hw)), "ndcg": [float(x) for x in nd]})
# reversed list: tau = -1
sw = list(range(10))
cases.append({"sw_idx": sw, "sw_val": [1.0 - 0.05 * i for i in range(10)], "hw_idx": sw[::-1], "hw_val": [1.0 - 0.05 * i for i in range(10)],
              "kendall": float(ns["kendall_tau"](sw, sw[::-1])),
              "ndcg": [float(x) for x in ns["ndcg"](sw, [1.0 - 0.05 * i for i in range(10)], sw[::-1], [1.0 - 0.05 * i for i in range(10)])]})
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "metrics_golden.json")
json.dump({"source": "plot_errors.py kendall_tau / ndcg executed from the reference checkout", "cases": cases}, open(out, "w"), indent=1)
print("wrote", out, len(cases), "cases")
