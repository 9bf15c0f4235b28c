#!/usr/bin/env python3
"""Condense rocprofv3's rocpd SQLite output (ROCm 7.2 default) into the small CSVs kept under profiles/.
    python tools/summarize_rocpd.py <tag> --stats <kt_results.db> [--pmc name=<p_results.db> ...] [--note "..."]
Writes profiles/<tag>_kernel_stats.csv (per-kernel calls / total / average / min / max duration, ns) and
profiles/<tag>_pmc_summary.csv (mean counter value per launch, per kernel)."""
import argparse
import csv
import os
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--stats")
ap.add_argument("--pmc", action="append", default=[])
ap.add_argument("--note", default="")
ap.add_argument("--only", default="step_kernel,reset_kernel,actor", help="kernel-name substrings kept in the PMC summary")
a = ap.parse_args()
os.makedirs("profiles", exist_ok=True)
if a.stats:
    c = sqlite3.connect(a.stats)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                     "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(f"profiles/{a.tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0][:100], r[1], r[2], f"{r[3]:.1f}", f"{100.0 * r[2] / tot:.2f}", r[4], r[5]])
if a.pmc:
    keep = [s for s in a.only.split(",") if s]
    with open(f"profiles/{a.tag}_pmc_summary.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["pass", "kernel", "counter", "mean_per_launch", "launches"])
        for spec in a.pmc:
            name, path = spec.split("=", 1)
            c = sqlite3.connect(path)
            q = ("select kernel_name, counter_name, avg(v), count(*) from (select kernel_name, counter_name, dispatch_id, "
                 "sum(value) as v from counters_collection group by kernel_name, counter_name, dispatch_id) "
                 "group by kernel_name, counter_name order by kernel_name, counter_name")
            for k, cn, v, n in c.execute(q):
                if any(s in k for s in keep):
                    w.writerow([name, k[:90], cn, v, n])
if a.note:
    open(f"profiles/{a.tag}_NOTE.txt", "w").write(a.note + "\n")
for suffix in ("kernel_stats", "pmc_summary"):
    p = f"profiles/{a.tag}_{suffix}.csv"
    if os.path.exists(p):
        print(open(p).read())
