#!/usr/bin/env python3
"""Condense rocprofv3's rocpd SQLite output (ROCm 7.2 default) into the small CSVs kept under profiles/.
    python tools/summarize_rocpd.py <tag> --stats <kt_results.db> [--pmc name=<p_results.db> ...] [--note "..."]
Writes profiles/<tag>_kernel_stats.csv (calls / total / average / min / max duration, ns) and
profiles/<tag>_pmc_summary.csv (mean counter value per launch).  Rows are grouped by (kernel, GRID SIZE in work-items):
one kernel launched at two batch sizes in the same run (bench.py's 524 288- and 2 097 152-env legs) gets one row per
size, so every figure quoted from a leg can be recomputed from its own row."""
import argparse
import csv
import os
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--stats")
ap.add_argument("--pmc", action="append", default=[])
ap.add_argument("--note", default="")
ap.add_argument("--only", default="step_kernel,reset_kernel,actor,calib", help="kernel-name substrings kept in the PMC summary")
ap.add_argument("--name-width", type=int, default=100)
ap.add_argument("--outdir", default="profiles", help="where the CSVs go (a GPU job writes under gpurun_out/ and the builder copies them)")
a = ap.parse_args()
os.makedirs(a.outdir, exist_ok=True)
if a.stats:
    c = sqlite3.connect(a.stats)
    rows = c.execute("select name, grid_x * grid_y * grid_z, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name, grid_x * grid_y * grid_z order by sum(duration) desc").fetchall()
    tot = sum(r[3] for r in rows) or 1
    with open(f"{a.outdir}/{a.tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "GridWorkItems", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0][:a.name_width], r[1], r[2], r[3], f"{r[4]:.1f}", f"{100.0 * r[3] / tot:.2f}", r[5], r[6]])
if a.pmc:
    keep = [s for s in a.only.split(",") if s]
    with open(f"{a.outdir}/{a.tag}_pmc_summary.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["pass", "kernel", "grid_work_items", "counter", "mean_per_launch", "launches"])
        for spec in a.pmc:
            name, path = spec.split("=", 1)
            c = sqlite3.connect(path)
            q = ("select kernel_name, grid_size, counter_name, avg(v), count(*) from (select kernel_name, grid_size, counter_name, "
                 "dispatch_id, sum(value) as v from counters_collection group by kernel_name, grid_size, counter_name, dispatch_id) "
                 "group by kernel_name, grid_size, counter_name order by kernel_name, grid_size, counter_name")
            for k, g, cn, v, n in c.execute(q):
                if any(s in k for s in keep):
                    w.writerow([name, k[:a.name_width], g, cn, v, n])
if a.note:
    open(f"{a.outdir}/{a.tag}_NOTE.txt", "w").write(a.note + "\n")
for suffix in ("kernel_stats", "pmc_summary"):
    p = f"{a.outdir}/{a.tag}_{suffix}.csv"
    if os.path.exists(p):
        print(open(p).read())
