#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/...) into small files under profiles/.
    python tools/summarize_prof.py <tag> --stats <kernel_stats.csv> [--pmc name=<counter_collection.csv> ...]
"""
import argparse
import collections
import csv
import os

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--stats")
ap.add_argument("--pmc", action="append", default=[])
ap.add_argument("--note", default="")
a = ap.parse_args()
os.makedirs("profiles", exist_ok=True)
if a.stats:
    rows = list(csv.reader(open(a.stats)))
    with open(f"profiles/{a.tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:100]
            w.writerow(r)
if a.pmc:
    with open(f"profiles/{a.tag}_pmc_summary.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["pass", "kernel", "counter", "mean_per_launch", "launches"])
        for spec in a.pmc:
            name, path = spec.split("=", 1)
            d = collections.defaultdict(list)
            for r in csv.DictReader(open(path)):
                if "step_kernel" in r["Kernel_Name"]:
                    d[(r["Kernel_Name"][:80], r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in sorted(d.items()):
                w.writerow([name, k, c, sum(v) / len(v), len(v)])
if a.note:
    open(f"profiles/{a.tag}_NOTE.txt", "w").write(a.note + "\n")
print(open(f"profiles/{a.tag}_pmc_summary.csv").read() if a.pmc else "ok")
