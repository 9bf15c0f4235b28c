"""Builds and drives the host simulation of the thread-per-env step kernel (tests/sanitize/tpe_host.cpp): the shipped device
headers compiled as host C++ against tests/sanitize/hip_shim, linked with the C oracle, under ASan + UBSan or MSan.

    python tests/sanitize/host_sim.py [asan|msan|plain] [--full]

Test infrastructure only (tests/test_sanitizers.py); nothing here is shipped or measured."""
import ctypes as C
import os
import struct
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
CSRC = os.path.join(ROOT, "marbler_amd", "csrc")
OUT = os.path.join(ROOT, "marbler_amd", "build", "host_sim")
CLANG_DIRS = ("/opt/rocm/lib/llvm/bin", "/opt/rocm/llvm/bin")
MODES = {
    # halt on the first report: -fno-sanitize-recover=all
    "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"],
    "msan": ["-fsanitize=memory", "-fsanitize-memory-track-origins=2", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"],
    "plain": [],
}
DEPS = [os.path.join(CSRC, h) for h in ("step_tpe.h", "device_common.h", "kernel_args.h", "sim_math.h")] + \
       [os.path.join(ROOT, "include", "robogym.h"), os.path.join(ROOT, "oracle", "oracle.c"), os.path.join(ROOT, "oracle", "oracle_core.h"),
        os.path.join(ROOT, "oracle", "oracle.h"), os.path.join(HERE, "tpe_host.cpp"), os.path.join(HERE, "tpe_host_inst.cpp"),
        os.path.join(HERE, "hip_shim", "hip", "hip_runtime.h"), os.path.abspath(__file__)]


def clang(cxx=True):
    for d in CLANG_DIRS:
        p = os.path.join(d, "clang++" if cxx else "clang")
        if os.path.exists(p):
            return p
    raise RuntimeError("ROCm's clang not found (the sanitizer runtimes of this image live under /opt/rocm/lib/llvm)")


def build(mode, rollout_n=(5, 7), opt="-O1", jobs=8):
    """-> path of the executable.  One translation unit per agent count, compiled side by side."""
    exe = os.path.join(OUT, f"tpe_host_{mode}")
    if os.path.exists(exe) and os.path.getmtime(exe) > max(os.path.getmtime(d) for d in DEPS):
        return exe
    os.makedirs(OUT, exist_ok=True)
    common = [opt, "-g", "-DRG_HOST_SIM", "-DRG_TPE_NO_W3", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-mavx2", "-mf16c",
              "-Wall", "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-unused-variable", "-Wno-unused-but-set-variable"] + MODES[mode]
    inc = ["-I", os.path.join(HERE, "hip_shim"), "-I", CSRC]
    jobs_ = []
    for n in range(2, 9):
        obj = os.path.join(OUT, f"inst_n{n}_{mode}.o")
        jobs_.append((obj, [clang(), "-std=c++17"] + common + inc + [f"-DSIM_N={n}"] + (["-DSIM_ROLLOUT"] if n in rollout_n else []) +
                      ["-c", os.path.join(HERE, "tpe_host_inst.cpp"), "-o", obj]))
    obj = os.path.join(OUT, f"main_{mode}.o")
    jobs_.append((obj, [clang(), "-std=c++17"] + common + inc + ["-c", os.path.join(HERE, "tpe_host.cpp"), "-o", obj]))
    obj = os.path.join(OUT, f"oracle_{mode}.o")
    jobs_.append((obj, [clang(cxx=False), "-std=c11"] + common + ["-c", os.path.join(ROOT, "oracle", "oracle.c"), "-o", obj]))

    def run(job):
        r = subprocess.run(job[1], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(" ".join(job[1]) + "\n" + r.stderr[-4000:])
        return job[0]

    with ThreadPoolExecutor(jobs) as pool:
        objs = list(pool.map(run, jobs_))
    subprocess.check_call([clang()] + MODES[mode] + ["-o", exe] + objs + ["-lpthread", "-lm"])
    return exe


# ---------------------------------------------------------------------------------------------- cases
def overrides_for(scenario, n, variant):
    """The overrides of one (scenario, agent count) cell; `variant` walks through the branches of the parameter block."""
    v = variant % 4
    common = {"penalize_violations": v != 3, "barrier_certificate": "default" if v == 2 else "safe",
              "collision_variant": "center" if v == 1 else "offset"}
    if scenario == "PredatorCapturePrey":
        npred = max(1, n // 2)
        ov = {"predator": npred, "capture": n - npred, "n_agents": n, "num_prey": [6, 9, 2, 6][v], "num_neighbors": [3, n + 1, 0, 2][v],
              "capability_aware": v == 1}
    elif scenario == "Warehouse":
        ov = {"n_agents": n, "num_neighbors": [5, 3, n, 7][v]}
    elif scenario == "MaterialTransport":
        nf = n // 2
        ov = {"n_agents": n, "n_fast_agents": nf, "n_slow_agents": n - nf, "start_dist": 0.3 if n <= 5 else 0.25, "capability_aware": v == 1}
    elif scenario == "Simple":
        ov = {"n_agents": n}
    else:
        return {k: common[k] for k in ("penalize_violations", "collision_variant")}
    ov.update(common)
    if v == 3:
        ov["update_frequency"] = 33      # 15 + 15 + 3: a remainder chunk
        ov["max_episode_steps"] = 7      # frequent resets
    return ov


def make_cases(full=False):
    """(scenario, overrides, n_act, E, K, rollout_k, time_limit) for every scenario x N in 2..8 (the instantiations of
    step_tpe.h), each with a ragged batch (two whole waves and a partly filled one), plus the multi-step form and the gymma
    block on a few."""
    cases = []
    for n in range(2, 9):
        for si, scenario in enumerate(["PredatorCapturePrey", "Warehouse", "MaterialTransport", "Simple", "ArcticTransport"]):
            if scenario == "MaterialTransport" and n < 4:
                continue
            if scenario == "ArcticTransport" and n != 4:
                continue
            n_act = 20 if scenario == "MaterialTransport" else 5
            for variant in range(4 if full else 2):
                var = variant + n + si if not full else variant
                ov = overrides_for(scenario, n, var)
                K = (10 if scenario == "MaterialTransport" else 16) * (2 if full else 1)
                cases.append((scenario, ov, n_act, 130 if variant % 2 == 0 else 67, K, 0, 9 if variant == 1 else 0))
    for n in (5, 7):   # rg_rollout's multi-step form (whole waves only, see tpe_host.cpp)
        for scenario in ("PredatorCapturePrey", "MaterialTransport", "Warehouse"):
            n_act = 20 if scenario == "MaterialTransport" else 5
            cases.append((scenario, overrides_for(scenario, n, 0), n_act, 128, 12, 4, 0))
    return cases


def write_cases(path, cases, seed=99):
    from helpers import oracle_reset_params
    from marbler_amd.params import load_config, make_params
    from oracle import c_oracle
    with open(path, "wb") as f:
        for i, (scenario, ov, n_act, E, K, rollout_k, time_limit) in enumerate(cases):
            cfg = load_config(scenario, None, ov)
            p = make_params(scenario, cfg)
            op = c_oracle.params_from_config(scenario, cfg, dtype=np.float32)
            rp = oracle_reset_params(c_oracle, p)
            a = np.random.RandomState(1000 + i).randint(0, n_act, size=(K, E, p.n_agents)).astype(np.int32)
            f.write(struct.pack("<I9iQq", 0x4D534752, C.sizeof(p), C.sizeof(op), C.sizeof(rp), E, K, 1, rollout_k, time_limit, 0, seed, 0))
            f.write(bytes(p))
            f.write(bytes(op))
            f.write(bytes(rp))
            f.write(a.tobytes())
    return len(cases)


def run(mode, full=False, timeout=1500, fault=None, only_first=False):
    """fault: 'uninit' / 'overrun' (SIM_FAULT, the harness's deliberate errors); only_first: the first case only."""
    exe = build(mode)
    cases = make_cases(full)[:1] if only_first else make_cases(full)
    cases_path = os.path.join(OUT, f"cases_{'full' if full else 'tier'}{'_first' if only_first else ''}.bin")
    n = write_cases(cases_path, cases)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:detect_stack_use_after_return=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", MSAN_OPTIONS="halt_on_error=1:abort_on_error=0")
    if fault:
        env["SIM_FAULT"] = fault
    else:
        env.pop("SIM_FAULT", None)
    r = subprocess.run([exe, cases_path], capture_output=True, text=True, timeout=timeout, env=env)
    return n, r


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "asan"
    n, r = run(mode, full="--full" in sys.argv)
    sys.stdout.write(r.stdout[-6000:])
    sys.stderr.write(r.stderr[-6000:])
    print(f"{mode}: {n} cases, exit code {r.returncode}")
    sys.exit(r.returncode)
