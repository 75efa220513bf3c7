#!/usr/bin/env python3
"""Condenses one tools/profile_round.sh run into the small files kept under profiles/:
  <tag>_kernel_stats.csv, <tag>_domain_stats.csv  (rocprofv3 --kernel-trace --stats, verbatim)
  <tag>_bench_plain.json, <tag>_bench_under_rocprofv3.json
  <tag>_multi{8,4}_kernel_stats.csv, <tag>_multi{8,4}_under_rocprofv3.json  (the multi-query path alone)
  <tag>_pmc_summary.json   per-launch means of the counters for the stream kernel
  pmc_traffic.json         HBM-side bytes per launch (FETCH_SIZE doubled: gfx950 correction of MI355X_MICROARCH.md)
Written next to the raw output (gpurun_out/prof_<tag>/summary/); copy that directory's content into profiles/."""
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)


def find(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    return hits[0] if hits else None


for src, name in ((find("ktrace/**/*kernel_stats.csv"), f"{tag}_kernel_stats.csv"),
                  (find("ktrace/**/*domain_stats.csv"), f"{tag}_domain_stats.csv")):
    if src:
        shutil.copy(src, os.path.join(dst, name))
for src, name in (("bench_plain.json", f"{tag}_bench_plain.json"),
                  ("bench_steps20.json", f"{tag}_bench_steps20.json"),
                  ("bench_under_rocprofv3.json", f"{tag}_bench_under_rocprofv3.json")):
    p = os.path.join(out, src)
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(dst, name), "w").write(lines[-1] + "\n")

for q in (8, 4):  # the multi-query path alone: per-kernel summary + the JSON line of that profiled run
    src = find(f"mtrace{q}/**/*kernel_stats.csv")
    if src:
        shutil.copy(src, os.path.join(dst, f"{tag}_multi{q}_kernel_stats.csv"))
    p = os.path.join(out, f"multi{q}.json")
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(dst, f"{tag}_multi{q}_under_rocprofv3.json"), "w").write(lines[-1] + "\n")

for sub, name in (("single", f"{tag}_single_query_kernel_stats.csv"), ("q17f", f"{tag}_config4_q17f32_kernel_stats.csv"),
                  ("q17f2", f"{tag}_config4_q17f32_two_chains_kernel_stats.csv")):
    src = find(f"{sub}/**/*kernel_stats.csv")
    if src:
        shutil.copy(src, os.path.join(dst, name))
    log = os.path.join(out, f"{sub}.log")
    if os.path.exists(log):
        shutil.copy(log, os.path.join(dst, name.replace("_kernel_stats.csv", ".log")))

for sub, stem in (("shard125k", f"{tag}_shard125k"), ("config3", f"{tag}_config3_10M")):
    src = find(f"{sub}/**/*kernel_stats.csv")
    if src:
        shutil.copy(src, os.path.join(dst, stem + "_kernel_stats.csv"))
    p = os.path.join(out, sub + ".json")
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(dst, stem + "_under_rocprofv3.json"), "w").write(lines[-1] + "\n")

summary = {}
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    # queries the profiled run executed: warm-up + timed steps + the repetitions of `timing` (bench.py --skip-warm)
    n_queries = None
    try:
        line = [l for l in open(d + ".json").read().splitlines() if l.startswith("{")][-1]
        j = json.loads(line)
        tm = j.get("timing", {})
        n_queries = j["warmup"] + j["steps"] + (tm.get("repetitions", 0) + tm.get("dropped", 0)) * tm.get("queries_per_repetition", 0)
    except Exception:
        pass
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            # the kernels that stream the matrix for a top-k query: batch_kernel (up to 32 queries per launch) and
            # stream_kernel<.., false, ..> (one query); Lb1E = the SpMV-only variant
            if not (("batch_kernel" in kn) or ("stream_kernel" in kn and "Lb1E" not in kn and "true" not in kn)):
                continue
            acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for name, per in acc.items():
            total = sum(per.values())
            summary[name] = {"launches": len(per), "queries": n_queries, "sum": total,
                             "mean_per_query": total / n_queries if n_queries else None}
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    fetch = summary["FETCH_SIZE"]["mean_per_query"] * 1024.0 * 2.0  # KiB -> B, x2: gfx950 tallies 128-B requests at 64 B
    write = summary["WRITE_SIZE"]["mean_per_query"] * 1024.0
    json.dump({"stream_kernel_hbm_bytes_per_launch": fetch + write, "unit": "bytes per query (a batch launch streams the matrix once per query)", "fetch_bytes_corrected": fetch, "write_bytes": write,
               "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --skip-warm, tag {tag}; "
                         "FETCH_SIZE x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE x 1024, summed over the launches and divided by the queries they served"},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: round(v["mean_per_query"] or 0, 2) for k, v in summary.items()}))
print(open(os.path.join(dst, f"{tag}_bench_plain.json")).read()[:600] if os.path.exists(os.path.join(dst, f"{tag}_bench_plain.json")) else "no plain bench line")
ks = os.path.join(dst, f"{tag}_kernel_stats.csv")
if os.path.exists(ks):
    print(open(ks).read()[:1500])
