"""Summary of tools/ladder.sh: medians per rung over the interleaved rounds, the steps between rungs, and the clocks the side
process sampled during the run.   python tools/ladder_summary.py DIR   (reads DIR/ladder.jsonl and DIR/clocks.csv)"""
import csv
import json
import os
import sys

import numpy as np


def main():
    d = sys.argv[1]
    runs = [json.loads(ln) for ln in open(os.path.join(d, "ladder.jsonl")) if ln.startswith("{")]
    names = []
    for r in runs:
        if r["name"] not in names:
            names.append(r["name"])
    print(f"attribution ladder, {runs[0]['rows']} x {runs[0]['cols']}, {runs[0]['nnz_per_row']} nnz/row, interleaved on one box; us per query")
    print(f"{'rung':16s} {'runs':>4s} {'sustained median':>17s} {'(min..max of runs)':>20s} {'p95/median':>11s} {'driver line (20 q)':>19s} {'load-only floor':>16s}")
    floors = [r["read_only_us"] for r in runs]
    prev = float(np.median(floors))
    print(f"{'L0 load-only':16s} {len(floors):4d} {prev:17.2f} {min(floors):9.2f}..{max(floors):<9.2f}")
    for n in names:
        rr = [r for r in runs if r["name"] == n]
        med = [r["sustained_median_us"] for r in rr]
        m = float(np.median(med))
        print(f"{n:16s} {len(rr):4d} {m:17.2f} {min(med):9.2f}..{max(med):<9.2f} {np.median([r['p95_over_median'] for r in rr]):11.3f} "
              f"{np.median([r['driver_line_kernel_us'] for r in rr]):19.2f} {np.median([r['read_only_us'] for r in rr]):16.2f}   step {m - prev:+.2f}")
        prev = m
    p = os.path.join(d, "clocks.csv")
    if os.path.exists(p):
        lines = open(p).read().split("\n")
        mine = None
        if lines and lines[0].startswith("#"):  # card list of the sampler: keep only the card the probes ran on
            pci = next((r.get("pci") for r in runs if r.get("pci")), None)
            for tok in lines[0][1:].split():
                c, _, addr = tok.partition("=")
                if pci and addr.lower() == pci.lower():
                    mine = c
            lines = lines[1:]
        rows = list(csv.DictReader(lines))
        if rows:
            if mine:
                print(f"\n(the probes ran on {mine} = {pci}: its columns only)")
                rows = [{k: v for k, v in r.items() if k == "t_s" or k.startswith(mine + "_")} for r in rows]
            print(f"\nclocks ({len(rows)} samples over {float(rows[-1]['t_s']):.0f} s, sysfs):")
            for key in rows[0]:
                if key == "t_s":
                    continue
                v = [float(r[key]) for r in rows if r.get(key) not in (None, "")]
                if v:
                    print(f"  {key:22s} min {min(v):8.1f}  p5 {np.percentile(v, 5):8.1f}  median {np.median(v):8.1f}  p95 {np.percentile(v, 95):8.1f}  max {max(v):8.1f}")


if __name__ == "__main__":
    main()
