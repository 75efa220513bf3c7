import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("value", round(d["value"]), "kernel_us", round(r["kernel_us"], 2), "frac", round(r["frac"], 3), "vs_read_only", round(r["read_only"]["headline_kernel_vs_read_only"], 3), "read_us", round(r["read_only"]["us_per_pass"], 2))
t = d["timing"]
print("timing: sustained median %.2f p95 %.2f (x%.3f) | one call per repetition median %.2f" % (t["kernel_us_median"], t["kernel_us_p95"], t["p95_over_median"], t["one_call_per_repetition"]["kernel_us_median"]))
print("thresholds", d["thresholds"]["checks_failed"], "nonstationary", {k: v for k, v in d["nonstationary"].items() if k not in ("note",)})
if "single_query" in d:
    s = d["single_query"]
    print("single", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in s.items() if k in ("kernel_us", "end_to_end_us", "frac", "counters", "parity_checked")})
    if s.get("native_loop"):
        print("single, native loop", {k: round(v, 2) for k, v in s["native_loop"].items() if isinstance(v, float)})
    print("cache_warm", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["cache_warm"].items() if k != "note"})
    for c in d["configs"]:
        print(c["workload"][:44], round(c.get("kernel_us", 0), 2), round(c.get("roofline", {}).get("frac", 0), 3), c.get("parity_checked"), c.get("error"))
    print("multi", [(x["queries_per_pass"], round(x["value"])) for x in d["multi_query"]["runs"]])
    print("kernels_us", {k: round(v, 2) for k, v in d["kernels_us"].items()})
print("traffic", r["traffic"], r["traffic_source"], "state_bytes", d.get("exchange_state_bytes"))
if "cpu_baseline" in d:
    print("cpu", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"])
