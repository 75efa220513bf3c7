#!/usr/bin/env python3
"""Sums the counters collected by tools/pmc_issue.sh per streaming kernel (all dispatches of the run) and derives a few
ratios: VALU / LDS / scalar / VMEM issue activity per wave cycle, LDS bank-conflict share, instructions per launch."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
res = {}
for d in sorted(glob.glob(os.path.join(out, "*"))):
    if not os.path.isdir(d):
        continue
    run = os.path.basename(d).rsplit("_", 1)[0] if os.path.basename(d).startswith("multi") else "batch"
    want = "multi_kernel" if run.startswith("multi") else "batch_kernel"
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        launches = set()
        for row in csv.DictReader(open(f)):
            if want not in row.get("Kernel_Name", ""):
                continue
            r = res.setdefault(run, {})
            r[row["Counter_Name"]] = r.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            launches.add(row["Dispatch_Id"])
        if launches:
            res.setdefault(run, {})["launches"] = len(launches)
# queries each profiled run served (from the JSON line it printed): counters per query
for run in list(res):
    n = None
    for f in sorted(glob.glob(os.path.join(out, ("batch" if run == "batch" else run + "_") + "*.json"))):
        try:
            j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
            n = j["warmup"] + j["steps"]
            if "timing" in j:  # the headline run repeats its timed batch (repetitions + 2 dropped ones)
                n += (j["timing"]["repetitions"] + j["timing"]["dropped"]) * j["timing"]["queries_per_repetition"]
                n += 8 * min(max(j["steps"], 32), 512)  # (timing.one_call_per_repetition: 10 calls, 2 of them not reported)... 10 calls in all
                n += 2 * min(max(j["steps"], 32), 512)
            if "nonstationary" in j:  # (round 4: the leg with queries of changing scale runs on the same engine)
                n += j["nonstationary"]["queries"] + 2 * min(max(j["steps"], 32), 512)  # (+ its two dropped repetitions)
            break
        except Exception:  # noqa: BLE001
            pass
    if n:
        res[run]["queries"] = n
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"):
            if k in res[run]:
                res[run][k + "_per_query"] = res[run][k] / n
# The headline kernel launch by launch (round 5: the profiled run is bench.py --headline-only, whose launches the line above does not
# count): per counter the MEDIAN launch of the kernel of local thresholds (32 queries each) / 32 -- and per packet of the matrix.
per = {}
for d in sorted(glob.glob(os.path.join(out, "batch*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            if "batch_kernel" not in kn or not (kn.rstrip(">)").split(",")[-1].strip().startswith("true") or ", true>" in kn):
                continue
            key = (row["Counter_Name"], row["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
if per and "batch" in res:
    med = {}
    for name in sorted({k[0] for k in per}):
        v = sorted(x for (n, _), x in per.items() if n == name)
        med[name] = v[len(v) // 2]
    n_packets = None
    for f in sorted(glob.glob(os.path.join(out, "batch*.json"))):
        try:
            j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
            n_packets = j.get("n_packets") or n_packets
        except Exception:  # noqa: BLE001
            pass
    n_packets = n_packets or 76404  # (BASELINE configs[1]: 19 501 236 nnz in packets of 256 entries, partitions padded to whole packets)
    res["batch"]["median_local_launch"] = {k: v for k, v in med.items()}
    res["batch"]["per_query_from_the_median_launch"] = {k: v / 32.0 for k, v in med.items()}
    res["batch"]["per_packet_from_the_median_launch"] = {k: v / 32.0 / n_packets for k, v in med.items() if k.startswith("SQ_INSTS")}
    res["batch"]["packets_per_query"] = n_packets
    for k in list(res["batch"]):  # (the per-query figures above divide by a query count the headline-only line does not carry)
        if k.endswith("_per_query") or k == "queries":
            del res["batch"][k]
for run, r in res.items():
    wc = r.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_ANY",
                  "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in r:
                r[k + "_per_wave_cycle"] = r[k] / wc
    if r.get("SQ_LDS_IDX_ACTIVE"):
        r["lds_bank_conflict_share"] = r.get("SQ_LDS_BANK_CONFLICT", 0.0) / r["SQ_LDS_IDX_ACTIVE"]
    if r.get("SQ_BUSY_CYCLES") and r.get("SQ_ACTIVE_INST_VALU"):
        r["valu_active_per_busy_cycle"] = r["SQ_ACTIVE_INST_VALU"] / r["SQ_BUSY_CYCLES"]
dst = os.path.join(os.path.dirname(os.path.abspath(out.rstrip("/"))), "..", "profiles", f"{tag}_pmc_issue.json")
dst = os.path.normpath(dst)
json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
os.makedirs(os.path.join(out, "summary"), exist_ok=True)  # gpurun brings back gpurun_out/ only: a copy travels with the raw files
json.dump(res, open(os.path.join(out, "summary", f"{tag}_pmc_issue.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
