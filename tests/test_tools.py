"""The evidence tooling (CPU tier): tools/summarize_rocpd.py must keep launches of one kernel at different batch sizes in
separate rows (the 524 288- and 2 097 152-env legs of the bench run were averaged into one), and
tools/ubench/hbm_calib.py must match kernels by their full name (a `calib_write1` row once swallowed `calib_write16`)."""
import csv
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_rocpd(path, with_counters):
    c = sqlite3.connect(path)
    c.execute("create table kernels (name text, grid_x int, grid_y int, grid_z int, duration int)")
    rows = [("void rg::tpe::step_kernel<0, 5, false>(rg::KernelArgs)", 524288, 1, 1, 160000 + i) for i in range(10)] + \
           [("void rg::tpe::step_kernel<0, 5, false>(rg::KernelArgs)", 2097152, 1, 1, 580000 + i) for i in range(10)] + \
           [("void rg::step_kernel<0, 8, false, 5, false, false>(rg::KernelArgs)", 65536, 1, 1, 15000 + i) for i in range(20)]
    c.executemany("insert into kernels values (?,?,?,?,?)", rows)
    if with_counters:
        c.execute("create table counters_collection (kernel_name text, grid_size int, counter_name text, dispatch_id int, value real)")
        rows = []
        for d in range(6):
            grid = 524288 if d < 3 else 2097152
            for inst in range(8):                       # one row per XCD instance: the tool sums them per dispatch
                rows.append(("void rg::tpe::step_kernel<0, 5, false>(rg::KernelArgs)", grid, "FETCH_SIZE", d, 10.0 if d < 3 else 40.0))
        c.executemany("insert into counters_collection values (?,?,?,?,?)", rows)
    c.commit()
    c.close()


def test_summarize_rocpd_keeps_batch_sizes_apart(tmp_path):
    kt, pm = str(tmp_path / "kt.db"), str(tmp_path / "p.db")
    _fake_rocpd(kt, False)
    _fake_rocpd(pm, True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_rocpd.py"), "t", "--outdir", str(tmp_path / "out"),
                        "--stats", kt, "--pmc", f"fetch={pm}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    stats = list(csv.DictReader(open(tmp_path / "out" / "t_kernel_stats.csv")))
    by = {(row["Name"], int(row["GridWorkItems"])): row for row in stats}
    assert len(by) == 3
    k = "void rg::tpe::step_kernel<0, 5, false>(rg::KernelArgs)"
    assert int(by[(k, 524288)]["Calls"]) == 10 and abs(float(by[(k, 524288)]["AverageNs"]) - 160004.5) < 1
    assert int(by[(k, 2097152)]["Calls"]) == 10 and abs(float(by[(k, 2097152)]["AverageNs"]) - 580004.5) < 1
    pmc = list(csv.DictReader(open(tmp_path / "out" / "t_pmc_summary.csv")))
    vals = {int(row["grid_work_items"]): (float(row["mean_per_launch"]), int(row["launches"])) for row in pmc}
    assert vals == {524288: (80.0, 3), 2097152: (320.0, 3)}      # 8 instances summed per dispatch, averaged per grid size


def test_calibration_summary_matches_kernels_by_their_full_name(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools", "ubench"))
    import hbm_calib
    db = str(tmp_path / "w.db")
    c = sqlite3.connect(db)
    c.execute("create table counters_collection (kernel_name text, grid_size int, counter_name text, dispatch_id int, value real)")
    n = hbm_calib.BYTES
    c.executemany("insert into counters_collection values (?,?,?,?,?)",
                  [("calib_write1(unsigned char*, unsigned long)", 256 * 16 * 256, "WRITE_SIZE", 0, 1.01 * n / 1024),
                   ("calib_write16(HIP_vector_type<float, 4u>*, unsigned long)", 256 * 16 * 256, "WRITE_SIZE", 1, n / 1024),
                   ("void calib_pose<true>(float*, int, int, float*)", (n // 60 + 3) // 4 * 64, "WRITE_SIZE", 2, (n // 60) * 60 / 1024)])
    c.commit()
    c.close()
    hbm_calib.summarize("t", [db], str(tmp_path))
    rows = list(csv.DictReader(open(tmp_path / "t_hbm_calibration.csv")))
    got = {(r["kernel"], r["pattern"]): float(r["reported_over_actual"]) for r in rows}
    assert got == {("calib_write1", "lane-contiguous"): 1.01, ("calib_write16", "lane-contiguous"): 1.0,
                   ("calib_pose<true>", "4 envs per wave"): 1.0}
