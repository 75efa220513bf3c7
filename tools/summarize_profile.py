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

# The headline kernel alone, launch by launch (the kernel trace of a --headline-only run): the first 100 launches of the kernel of
# local thresholds are tkspmv_create's pacing measurement (synthetic queries, six settings), left out; the rest are warm-up, timed
# region and repetitions -- 32 real queries each.
kt = find("ktrace/**/*kernel_trace.csv")
if kt:
    rows = [r for r in csv.DictReader(open(kt)) if "batch_kernel" in r.get("Kernel_Name", "")]
    def is_local(n):
        return n.rstrip(">)").split(",")[-1].strip().startswith("true") or ", true>" in n
    loc = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if is_local(r["Kernel_Name"]))
    exa = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if not is_local(r["Kernel_Name"])]
    line = {}
    p = os.path.join(out, "bench_under_rocprofv3.json")
    if os.path.exists(p):
        ls = [l for l in open(p).read().splitlines() if l.startswith("{")]
        line = json.loads(ls[-1]) if ls else {}
    skip = int(line.get("launches_of_the_pacing_measurement", 100))
    kept = [e - s for s, e in loc[skip:]]
    full = sorted(d for d in kept if d > 0.6 * sorted(kept)[len(kept) // 2]) if kept else []  # (the warm-up's and the region's last launches may be shorter)
    if full:
        json.dump({"kernel": "tkspmv::batch_kernel<4,1024,7,false,true>", "source": "rocprofv3 --kernel-trace of bench.py --headline-only (this tag's bench_under_rocprofv3.json)",
                   "launches_in_trace": len(loc), "launches_left_out": skip, "why": "tkspmv_create measures its pacing with them (synthetic queries)",
                   "full_launches": len(full), "queries_per_launch": 32, "mean_launch_us": sum(full) / len(full) / 1e3,
                   "median_launch_us": full[len(full) // 2] / 1e3, "p95_launch_us": full[int(0.95 * (len(full) - 1))] / 1e3,
                   "mean_us_per_query": sum(full) / len(full) / 1e3 / 32, "median_us_per_query": full[len(full) // 2] / 1e3 / 32,
                   "exact_kernel_launches": len(exa), "exact_kernel_mean_us": (sum(exa) / len(exa) / 1e3) if exa else None,
                   "bench_line_of_the_same_run": {k: line.get(k) for k in ("kernel_us", "sustained_median_us", "p95_over_median", "read_only_us", "pacing", "checks_failed")}},
                  open(os.path.join(dst, f"{tag}_headline_launches.json"), "w"), indent=1)

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

# configs[3] (10M rows): the launches of 32 queries alone -- tkspmv_create's pacing measurement launches 4 queries at a time there,
# and the kernel statistics above average over both kinds
kt3 = find("config3/**/*kernel_trace.csv")
if kt3:
    rows3 = [r for r in csv.DictReader(open(kt3)) if "batch_kernel" in r.get("Kernel_Name", "")]
    dur = {True: [], False: []}
    for r in rows3:
        n = r["Kernel_Name"]
        dur[n.rstrip(">)").split(",")[-1].strip().startswith("true") or ", true>" in n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out3 = {}
    for local, name in ((True, "checked local thresholds (batch_kernel<..., true>)"), (False, "device-wide exchange (batch_kernel<..., false>)")):
        d3 = sorted(dur[local])
        full = [x for x in d3 if x > 0.6 * d3[-1]] if d3 else []
        if full and full[-1] > 2e6:  # (launches of 32 queries at 10M rows take milliseconds: the empty repair launches do not count)
            out3[name] = {"launches_of_32_queries": len(full), "mean_launch_us": sum(full) / len(full) / 1e3, "median_launch_us": full[len(full) // 2] / 1e3,
                          "mean_us_per_query": sum(full) / len(full) / 1e3 / 32, "median_us_per_query": full[len(full) // 2] / 1e3 / 32, "other_launches_of_that_kernel": len(d3) - len(full)}
    if out3:
        json.dump(out3, open(os.path.join(dst, f"{tag}_config3_10M_launches.json"), "w"), indent=1)

for rows in (125000, 250000, 500000):  # the one-GPU rehearsal of a strong-scaled run's shards; the size sweep
    p = os.path.join(out, f"shard_rehearsal_{rows}.json")
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(dst, f"{tag}_shard_rehearsal_{rows}.json"), "w").write(lines[-1] + "\n")
if os.path.exists(os.path.join(out, "size_sweep.txt")):
    shutil.copy(os.path.join(out, "size_sweep.txt"), os.path.join(dst, f"{tag}_size_sweep.txt"))

summary = {}
QUERIES_PER_LAUNCH = 32  # bench.py's profiled runs use whole launches of the batch kernel (--steps / --warmup multiples of 32)
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        # Per launch and kernel: the counter summed over its rows (one per XCD / instance). Two kernels stream the matrix for the
        # headline: batch_kernel<..., true> (local thresholds: one pass per query) and, behind every such launch, batch_kernel<...,
        # false> with repair = 1 (empty unless a check failed). A run also holds other launches of these kernels (short batches of
        # the warm-up, the nonstationary leg with its repairs), so the per-query figure is the MEDIAN full launch / 32, plus the
        # median launch of the exact kernel / 32 -- not a sum over the run divided by a query count reconstructed from the line.
        per = {}
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            if "batch_kernel" not in kn:
                continue
            which = "local" if kn.rstrip(">)").split(",")[-1].strip().startswith("true") or ", true>" in kn else "exact"
            key = (row["Counter_Name"], which, row["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
        for name in sorted({k[0] for k in per}):
            loc = sorted(v for (n, w, _), v in per.items() if n == name and w == "local")
            exa = sorted(v for (n, w, _), v in per.items() if n == name and w == "exact")
            if not loc:
                continue
            med_loc = loc[len(loc) // 2]
            med_exa = exa[len(exa) // 2] if exa else 0.0
            summary[name] = {"local_launches": len(loc), "exact_launches": len(exa), "median_per_local_launch": med_loc,
                             "median_per_exact_launch": med_exa, "queries_per_launch": QUERIES_PER_LAUNCH,
                             "mean_per_query": (med_loc + med_exa) / QUERIES_PER_LAUNCH}
if summary:  # (a run of part B only has no counter passes: leave part A's file alone)
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    fetch = summary["FETCH_SIZE"]["mean_per_query"] * 1024.0 * 2.0  # KiB -> B, x2: gfx950 tallies 128-B requests at 64 B
    write = summary["WRITE_SIZE"]["mean_per_query"] * 1024.0
    json.dump({"stream_kernel_hbm_bytes_per_launch": fetch + write, "unit": "bytes per query (a batch launch streams the matrix once per query)", "fetch_bytes_corrected": fetch, "write_bytes": write,
               "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --skip-warm, tag {tag}; "
                         "FETCH_SIZE x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE x 1024: the median 32-query launch of the "
                         "batch kernel plus the median (empty) launch of the exact kernel behind it, divided by 32"},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: round(v["mean_per_query"] or 0, 2) for k, v in summary.items()}))
print(open(os.path.join(dst, f"{tag}_bench_plain.json")).read()[:600] if os.path.exists(os.path.join(dst, f"{tag}_bench_plain.json")) else "no plain bench line")
ks = os.path.join(dst, f"{tag}_kernel_stats.csv")
if os.path.exists(ks):
    print(open(ks).read()[:1500])
