#!/usr/bin/env python3
"""Driver of tools/ubench/hbm_calib.hip: launches every calibration kernel a few times on a buffer past the Infinity
Cache and prints the bytes each launch touches (JSON, one line per kernel).  Run it under rocprofv3 to get the counters:

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/calib_f -o p -- python3 tools/ubench/hbm_calib.py
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/calib_w -o p -- python3 tools/ubench/hbm_calib.py
    python tools/ubench/hbm_calib.py --summarize gpurun_out/calib_f/.../p_results.db gpurun_out/calib_w/.../p_results.db

--summarize joins the counter databases with the known byte counts and writes profiles/<tag>_hbm_calibration.csv:
reported KB x 1024 / actual bytes per (kernel, grid), the factor DESIGN.md section 4.4 applies."""
import argparse
import ctypes
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
MODES = [(0, "calib_read4", 0), (1, "calib_read16", 0), (2, "calib_write4", 0), (3, "calib_write16", 0), (6, "calib_write1", 0),
         (4, "calib_pose<false>", 4), (4, "calib_pose<false>", 8), (5, "calib_pose<true>", 4), (5, "calib_pose<true>", 8)]
BYTES = 768 << 20           # 768 MiB: three times the Infinity Cache


def expected():
    """(kernel substring, grid work-items, bytes touched per launch, direction)"""
    rows = []
    for mode, name, epw in MODES:
        if mode in (4, 5):
            E = BYTES // 60
            grid = (E + epw - 1) // epw * 64
            rows.append((name, grid, E * 60, "write" if mode == 5 else "read", f"{epw} envs per wave"))
        else:
            unit = {0: 4, 1: 16, 2: 4, 3: 16, 6: 1}[mode]
            rows.append((name, 256 * 16 * 256, BYTES // unit * unit, "read" if mode < 2 else "write", "lane-contiguous"))
    return rows


def run():
    import torch
    lib = ctypes.CDLL(os.path.join(HERE, "libhbm_calib.so"))
    lib.calib_run.restype = ctypes.c_longlong
    lib.calib_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    buf = torch.zeros(BYTES, dtype=torch.uint8, device="cuda")
    other = torch.zeros(BYTES, dtype=torch.uint8, device="cuda")      # touched between launches: evicts `buf` from the caches
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for mode, name, epw in MODES:
        for rep in range(4):
            other.add_(1)
            torch.cuda.synchronize()
            n = lib.calib_run(mode, buf.data_ptr(), BYTES, epw, sink.data_ptr(), stream)
            torch.cuda.synchronize()
        print(json.dumps({"kernel": name, "envs_per_wave": epw, "bytes_per_launch": int(n)}), flush=True)
    assert float(sink.sum()) == 0.0   # the read kernels' "never true" sink stayed untouched


def summarize(tag, dbs, outdir=None):
    import csv
    import sqlite3
    exp = expected()
    outdir = outdir or os.path.join(os.path.dirname(os.path.dirname(HERE)), "profiles")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, f"{tag}_hbm_calibration.csv")
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "pattern", "grid_work_items", "direction", "actual_bytes_per_launch", "counter", "reported_KB_per_launch",
                    "reported_over_actual", "launches"])
        for path in dbs:
            c = sqlite3.connect(path)
            q = ("select kernel_name, grid_size, counter_name, avg(v), count(*) from (select kernel_name, grid_size, counter_name, "
                 "dispatch_id, sum(value) as v from counters_collection group by kernel_name, grid_size, counter_name, dispatch_id) "
                 "group by kernel_name, grid_size, counter_name")
            for k, g, cn, v, n in c.execute(q):
                for name, grid, nbytes, direction, pattern in exp:
                    if (name + "(") in (k.replace("void ", "") + "(") and k.replace("void ", "").startswith(name) and grid == g and ((cn == "FETCH_SIZE") == (direction == "read")) and cn in ("FETCH_SIZE", "WRITE_SIZE"):
                        w.writerow([name, pattern, g, direction, nbytes, cn, f"{v:.1f}", f"{v * 1024.0 / nbytes:.4f}", n])
    print(open(out).read())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--summarize", nargs="+")
    ap.add_argument("--tag", default="r3")
    ap.add_argument("--outdir", default=None)
    a = ap.parse_args()
    if a.summarize:
        summarize(a.tag, a.summarize, a.outdir)
    else:
        run()
