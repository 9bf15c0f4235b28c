#!/usr/bin/env python3
"""bench.py -- env agent-steps/sec of the HIP step engine (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no torchrun around it launches its own N ranks (one process per
GPU, backend nccl = RCCL; the parent never touches a GPU) and relays rank 0's line.

One "step" = one rg_step launch = one env step (U = 29 sim sub-iterations) of every env of
the workload: PredatorCapturePrey-v0, 4096 envs x 5 agents per GPU (BASELINE.json configs[1];
N GPUs = N x 4096 envs, weak scaling), random policy.  Actions are synthetic (uniform ints,
seed 1234 + rank) and resident in HBM before the timed region; finished envs are reset inside
the launch.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
PCP_OVERRIDES = {"predator": 3, "capture": 2, "n_agents": 5}   # BASELINE: 5 agents
# algorithmic bytes per env-step of a fully fused PCP step (N=5, D=16, P=6)
ALGO_BYTES_PER_ENV_STEP = 585
# the same for the other BASELINE configurations (Warehouse N=8, D=18; MaterialTransport N=6, D=9)
ALGO_BYTES_OTHER = {"Warehouse": 896, "MaterialTransport": 510}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS, BF16_MFMA_PEAK_TFLOPS = 157.3, 2500.0   # dense matrix peaks (MI355X_MICROARCH.md; never the 2:1-sparsity figures)


PMC_PROFILE = os.path.join("profiles", "r5_final_pmc_summary.csv")   # newest committed rocprofv3 --pmc passes of this command
PMC_FALLBACKS = (os.path.join("profiles", "r4_final_pmc_summary.csv"), os.path.join("profiles", "r3_final_pmc_summary.csv"))
HEADLINE_KERNEL = "rg::step_kernel<0, 8, false, 5, false"   # <PCP, GW 8, step, N 5, single launch[, no gymma block]>
HEADLINE_GRID = 1024 * 64                                     # 4096 envs at 4 per wavefront: 1024 one-wave workgroups
# rocprofv3's FETCH_SIZE on gfx950 reports half the bytes read, in every access shape of the step kernels (4 B per lane,
# 16 B per lane, the strided 60-byte pose blocks); WRITE_SIZE is exact (1.01 for 1-byte flags): measured on known byte
# counts past the Infinity Cache, tools/ubench/hbm_calib.py -> profiles/r3_hbm_calibration.csv
FETCH_SIZE_SCALE, WRITE_SIZE_SCALE = 2.0, 1.0
SIMDS, CLOCK_HZ, VALU_ISSUE_CYCLES = 1024, 2.4e9, 4     # 256 CUs x 4 SIMDs; MI355X_MICROARCH.md: 2400 MHz, one wave issues a VALU op per 4 cycles


def bench_overrides(scenario):
    """Config overrides of the BASELINE.json configurations: PCP 5 agents (configs[1], [3]),
    Warehouse 8 agents (configs[2]), MaterialTransport 6 heterogeneous agents (configs[4])."""
    if scenario == "PredatorCapturePrey":
        return dict(PCP_OVERRIDES)
    if scenario == "Warehouse":
        return {"n_agents": 8}
    if scenario == "MaterialTransport":
        return {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}
    return {}


def committed_counters():
    """Per-launch counters of the headline kernel from the COMMITTED rocprofv3 PMC passes of this same command
    (separate --pmc runs; FETCH_SIZE / WRITE_SIZE in KB).  Counters cannot be read from inside an un-profiled
    run: these are constants of the committed profile, labelled as such in the line.  Returns (dict, path)."""
    import csv
    for rel in (PMC_PROFILE,) + PMC_FALLBACKS:
        try:
            vals = {}
            for r in csv.DictReader(open(os.path.join(ROOT, rel))):
                if HEADLINE_KERNEL in r["kernel"] and int(r.get("grid_work_items") or HEADLINE_GRID) == HEADLINE_GRID:
                    vals[r["counter"]] = float(r["mean_per_launch"])
            if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
                return vals, rel
        except OSError:
            continue
    return {}, None


def live_counters(args, timeout=150):
    """The headline kernel's counters MEASURED in this run: three child passes of this same script under `rocprofv3 --pmc`
    (counters only -- no trace domains --, one small group per pass, as MI355X_MICROARCH.md prescribes), started while this process
    has not touched the GPU; each child times nothing, it only launches the headline workload.  Returns ({counter: mean per
    launch of the headline kernel at the headline grid}, description) or ({}, reason) -- the caller falls back to the committed
    profile (`committed_counters`) and says so.  RG_BENCH_NO_LIVE_COUNTERS=1 skips it."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if os.environ.get("RG_BENCH_NO_LIVE_COUNTERS"):
        return {}, "skipped (RG_BENCH_NO_LIVE_COUNTERS)"
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {}, "rocprofv3 not found"
    vals = {}
    child = [sys.executable, os.path.abspath(__file__), "--counters-child", "--steps", "200", "--warmup", "20", "--spinup-ms", "0"]
    try:
        for group in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_WAVES"]):
            tmp = tempfile.mkdtemp(prefix="rg_pmc_")
            try:
                env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
                r = subprocess.run([rocprof, "--pmc"] + group + ["-d", tmp, "-o", "p", "--"] + child, cwd="/tmp", env=env,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
                dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
                if r.returncode != 0 or not dbs:
                    return {}, f"rocprofv3 --pmc {' '.join(group)} failed (rc {r.returncode})"
                c = sqlite3.connect(dbs[0])
                q = ("select counter_name, avg(v), count(*) from (select counter_name, dispatch_id, sum(value) as v from counters_collection "
                     "where kernel_name like ? and grid_size = ? group by counter_name, dispatch_id) group by counter_name")
                for name, v, n in c.execute(q, ("%" + HEADLINE_KERNEL + "%", HEADLINE_GRID)):
                    if n >= 50:
                        vals[name] = float(v)
                c.close()
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
    except Exception as exc:   # a profiler problem must not cost the bench line
        return {}, f"live counter passes failed: {type(exc).__name__}: {exc}"
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return {}, "the counter passes returned no rows for the headline kernel"
    return vals, "measured in this run: rocprofv3 --pmc child passes of this script (FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU SQ_WAVES, 220 launches each)"


def counters_child(args):
    """What a `rocprofv3 --pmc` pass of live_counters() runs: the headline workload's launches, nothing timed or printed."""
    import torch
    from marbler_amd import VecRobotariumEnv
    dev = torch.device("cuda", 0)
    env = VecRobotariumEnv(args.scenario, args.envs_per_gpu, overrides=bench_overrides(args.scenario), device=dev, seed=0, auto_reset=True)
    n_act = 20 if args.scenario == "MaterialTransport" else 5
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    actions = torch.randint(0, n_act, (64, args.envs_per_gpu, env.N), generator=gen, device=dev, dtype=torch.int32)
    env.reset()
    for i in range(args.steps + args.warmup):
        env.step_raw(actions[i % 64].data_ptr())
    torch.cuda.synchronize(dev)
    env.close()


def _port_worker(seconds, seed):
    """One NumPy-port env stepped for `seconds` on this process's core; prints the env-step count."""
    import numpy as np
    from oracle import np_port
    from marbler_amd.params import load_config
    cfg = load_config("PredatorCapturePrey", overrides=dict(PCP_OVERRIDES, seed=seed))
    port = np_port.make_port("PredatorCapturePrey", cfg)
    rng = np.random.RandomState(seed)
    port.reset()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(25):
            _, _, d, _ = port.step(list(rng.randint(0, 5, size=5)))
            if d[0]:
                port.reset()
        n += 25
    print(f"PORT_STEPS {n} {time.perf_counter() - t0:.3f}")


def _oracle_worker(seconds, seed):
    """The C oracle (float64, 512 envs batched) stepped for `seconds` on this process's core."""
    import numpy as np
    from oracle import np_port
    from oracle.c_oracle import OracleVecEnv
    from marbler_amd.params import load_config
    cfg = load_config("PredatorCapturePrey", overrides=dict(PCP_OVERRIDES, seed=seed))
    port = np_port.make_port("PredatorCapturePrey", cfg)
    E = 512
    env = OracleVecEnv("PredatorCapturePrey", cfg, E, dtype=np.float64)
    for e in range(E):
        port.reset()
        env.set_state(e, poses=port.agent_poses, prey_loc=port.prey_loc)
    rng = np.random.RandomState(seed)
    acts = rng.randint(0, 5, size=(8, E, 5)).astype(np.int32)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        env.step(acts[n % 8])
        n += 1
    print(f"PORT_STEPS {n * E} {time.perf_counter() - t0:.3f}")


def usable_cores():
    """(cores this job may run on, why): the affinity mask, cut down by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0))
    why = f"sched_getaffinity ({n} of {os.cpu_count()} host cores)"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            q = max(1, int(float(quota) / float(period) + 0.5))
            if q < n:
                n, why = q, f"cgroup cpu.max = {quota}/{period} ({q} of {os.cpu_count()} host cores)"
    except (OSError, ValueError):
        pass
    return n, why


def cpu_all_cores(seconds=8.0, worker="_port_worker", n=None):
    """the same port, one env per process, one process per core (the shape of
    EPyMARL's parallel runner).  CPU-only child processes; returns (agent-steps/s, processes)."""
    import subprocess
    n = max(1, n or usable_cores()[0])
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; bench.{worker}({seconds}, int(sys.argv[1]))"
    procs = [subprocess.Popen([sys.executable, "-c", code, str(100 + i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              text=True, env=env) for i in range(n)]
    rate = 0.0
    for pr in procs:
        out, _ = pr.communicate(timeout=seconds * 6 + 120)
        for line in out.splitlines():
            if line.startswith("PORT_STEPS"):
                _, steps, dt = line.split()
                rate += int(steps) * 5 / float(dt)
    return rate, n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds_budget=12.0):
    """The reference-shaped NumPy port (oracle/np_port.py: one env per object, Python loop over
    the 29 sub-iterations, float64) on ONE host core, same scenario config, random policy.
    Also the C oracle (float64, one core) as the 'strong' CPU number."""
    import numpy as np
    from oracle import np_port
    from oracle.c_oracle import OracleVecEnv
    from marbler_amd.params import load_config
    cfg = load_config("PredatorCapturePrey", overrides=dict(PCP_OVERRIDES, seed=7))
    port = np_port.make_port("PredatorCapturePrey", cfg)
    rng = np.random.RandomState(1234)
    port.reset()
    for _ in range(20):
        _, _, d, _ = port.step(list(rng.randint(0, 5, size=5)))
        if d[0]:
            port.reset()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_budget:
        for _ in range(50):
            _, _, d, _ = port.step(list(rng.randint(0, 5, size=5)))
            if d[0]:
                port.reset()
        n += 50
    dt = time.perf_counter() - t0
    py_rate = n * 5 / dt
    # C oracle, batched, 1 core
    E = 512
    env = OracleVecEnv("PredatorCapturePrey", cfg, E, dtype=np.float64)
    for e in range(E):
        port.reset()
        env.set_state(e, poses=port.agent_poses, prey_loc=port.prey_loc)
    acts = rng.randint(0, 5, size=(40, E, 5)).astype(np.int32)
    t0 = time.perf_counter()
    for k in range(40):
        env.step(acts[k])
    c_rate = 40 * E * 5 / (time.perf_counter() - t0)
    # the same two CPU programmes with the barrier QP solved as the reference's stack solves it (the restated cvxopt iterate,
    # oracle/rps_restated/cvxopt_restated.py / oracle_core.h barrier_qp_ipm): the closest thing to the reference's real cost
    # per step that can be timed here -- real cvxopt is C behind Python, so the truth lies between these two figures
    cfg_ip = dict(cfg, barrier_solver="cvxopt")
    port_ip = np_port.make_port("PredatorCapturePrey", cfg_ip)
    port_ip.reset()
    n_ip, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 4.0:
        for _ in range(10):
            _, _, d, _ = port_ip.step(list(rng.randint(0, 5, size=5)))
            if d[0]:
                port_ip.reset()
        n_ip += 10
    py_ip_rate = n_ip * 5 / (time.perf_counter() - t0)
    np_port.make_port("PredatorCapturePrey", cfg)  # the solver switch is module state: back to the default
    env_ip = OracleVecEnv("PredatorCapturePrey", cfg_ip, E, dtype=np.float64)
    for e in range(E):
        port.reset()
        env_ip.set_state(e, poses=port.agent_poses, prey_loc=port.prey_loc)
    t0 = time.perf_counter()
    for k in range(10):
        env_ip.step(acts[k])
    c_ip_rate = 10 * E * 5 / (time.perf_counter() - t0)
    # under rocprofv3 the profiler's preloaded library has already initialised the GPU in this process:
    # no child processes then (each would be an exec after GPU initialisation)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith("ROCPROF") for k in os.environ)
    # every core this job may use (affinity mask / cgroup quota -- stated in the line), and the 16-core share of
    # one GPU of an 8-GPU, 128-core-per-socket host beside it
    n_all, why = usable_cores()
    n_all = min(n_all, 256)
    all_rate, n_proc = (None, 0) if profiled else cpu_all_cores(n=n_all)
    c_all, _ = (None, 0) if profiled else cpu_all_cores(4.0, "_oracle_worker", n=n_all)
    share = min(16, n_all)
    if profiled or share == n_all:
        share_rate, c_share = all_rate, c_all
    else:
        share_rate, _ = cpu_all_cores(n=share)
        c_share, _ = cpu_all_cores(4.0, "_oracle_worker", n=share)
    return {"value": py_rate, "unit": "agent-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} env-steps of 1 env x 5 agents, NumPy float64 port in the reference's shape "
                      f"(oracle/np_port.py), {dt:.1f} s on one core",
            "cpu_model": cpu_model(), "host_cores": os.cpu_count(), "usable_cores": n_all, "usable_cores_source": why,
            "all_cores": {"value": all_rate, "cores": n_proc, "kind": "port",
                          "sample": f"{n_proc} processes x 1 env each (EPyMARL parallel-runner shape) on every core this "
                                    f"job may use, 8 s" if not profiled else "skipped under rocprofv3 (no child processes)"},
            "gpu_share_16_cores": {"value": share_rate, "cores": share, "kind": "port"},
            "c_oracle_f64_1core": c_rate, "c_oracle_f64_all_cores": {"value": c_all, "cores": n_proc},
            "c_oracle_f64_gpu_share": {"value": c_share, "cores": share},
            "interior_point_mode_1core": {"port": py_ip_rate, "c_oracle_f64": c_ip_rate, "unit": "agent-steps/s",
                                          "sample": f"{n_ip} env-steps of the NumPy port (4 s) / 10 x {E} env-steps of the C "
                                                    f"oracle, barrier_solver: cvxopt (restated iterate, rps' options)"}}


SATURATED_ENVS = 524288
ROLLOUT_STEPS_PER_LAUNCH = 64


def rollout_leg(env, dev, n_act, seed, launches=16):
    """The same envs driven through rg_rollout: 64 env steps per launch for a pre-generated action
    sequence (what a random-policy rollout is), every step's outputs written to [64, ...] buffers.
    Envs advance independently inside the launch, so a launch costs 64 mean steps instead of 64
    slowest-wave steps.  Reported beside the headline (which stays one rg_step launch per step, the
    call a policy-in-the-loop trainer makes)."""
    import torch
    K = ROLLOUT_STEPS_PER_LAUNCH
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    acts = torch.randint(0, n_act, (K, env.E, env.N), generator=gen, device=dev, dtype=torch.int32)
    buf = env.rollout(acts)
    env.rollout(acts, out=buf)
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(launches):
        env.rollout(acts, out=buf)
    b.record()
    torch.cuda.synchronize(dev)
    ms = a.elapsed_time(b) / (launches * K)
    gbs = ALGO_BYTES_PER_ENV_STEP * env.E / (ms * 1e-3) / 1e9
    return {"api": "rg_rollout", "steps_per_launch": K, "launches": launches, "ms_per_step": ms,
            "agent_steps_per_s": env.E * env.N / (ms * 1e-3), "hbm_achieved_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}


def host_boundary_leg(env, dev, n_act, steps=200):
    """Side measurement (not `value`): the same step with a HOST consumer on both sides -- the rate a trainer sees that keeps its
    actions and transition data in host memory (the reference's own shape: NumPy in, NumPy out).  Per step: the action batch
    [E, N] int32 goes up from pinned memory, rg_step runs, observations / rewards / done flags come down to pinned memory, and
    the host waits for them (it needs them to choose the next actions).  The C ABI itself takes device pointers; `value` is
    measured with everything resident in HBM.  PCIe-inclusive by construction."""
    import numpy as np
    import torch
    E, N = env.E, env.N
    rng = np.random.RandomState(3)
    act_host = torch.from_numpy(rng.randint(0, n_act, size=(8, E, N)).astype(np.int32)).pin_memory()
    act_dev = torch.empty(E, N, dtype=torch.int32, device=dev)
    obs_h = torch.empty_like(env.obs, device="cpu").pin_memory()
    rew_h = torch.empty_like(env.reward, device="cpu").pin_memory()
    done_h = torch.empty_like(env.done_u8, device="cpu").pin_memory()
    nbytes = act_dev.numel() * 4 + obs_h.numel() * 4 + rew_h.numel() * 4 + done_h.numel()

    def one(i):
        act_dev.copy_(act_host[i % 8], non_blocking=True)
        env.step_raw(act_dev.data_ptr())
        obs_h.copy_(env.obs, non_blocking=True)
        rew_h.copy_(env.reward, non_blocking=True)
        done_h.copy_(env.done_u8, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()

    env._sync_stream()
    for i in range(20):
        one(i)
    t0 = time.perf_counter()
    for i in range(steps):
        one(i)
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"what": "rg_step with host-resident actions and outputs (pinned buffers, one synchronisation per step)", "steps": steps,
            "ms_per_step": ms, "agent_steps_per_s": E * N / (ms * 1e-3), "bytes_over_pcie_per_step": nbytes,
            "pcie_GBs": nbytes / (ms * 1e-3) / 1e9}


def graph_leg(env, dev, ptrs, K, regions=40):
    """The timed region's K rg_step launches recorded once into a hipGraph and replayed: the same kernels, no host call
    per step.  A 20-step region launched step by step is exposed to the host's jitter (1 region in 8 ran 10-20 % slow on
    the GPU box, tools/graph_region_probe.py); replayed as a graph its median is ~3 % lower and the tail goes away.
    Reported beside the headline, which stays one host call per step (what a policy-in-the-loop trainer without a
    captured loop makes).  Any failure here is reported as such and never touches the line's value."""
    import torch
    try:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        prev = env._stream
        env.set_stream(side)
        try:
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    for i in range(K):
                        env.step_raw(ptrs[i % len(ptrs)])
                graph.replay()
                torch.cuda.synchronize(dev)
                t = []
                for _ in range(regions):
                    t0 = time.perf_counter()
                    graph.replay()
                    torch.cuda.synchronize(dev)
                    t.append((time.perf_counter() - t0) / K)
        finally:
            env.set_stream(prev)
            torch.cuda.current_stream(dev).wait_stream(side)
        t.sort()
        med = t[len(t) // 2]
        return {"api": f"one hipGraph of {K} rg_step launches", "regions": regions, "ms_per_step_median": med * 1e3,
                "ms_per_step_min": t[0] * 1e3, "ms_per_step_max": t[-1] * 1e3, "agent_steps_per_s_median": env.E * env.N / med}
    except Exception as exc:   # noqa: BLE001 - a side measurement
        return {"error": repr(exc)[:200]}


def saturated_leg(dev, overrides, E=SATURATED_ENVS):
    """The same step at a batch that fills the chip (524288 envs = 8 waves per SIMD, and 2097152 = 32, where the
    ragged end of the launch weighs less; thread-per-env kernel): where the path stands against the HBM roof
    when launch latency no longer binds.  Reported beside the headline, never as `value`."""
    import torch
    from marbler_amd import VecRobotariumEnv
    env = VecRobotariumEnv("PredatorCapturePrey", E, overrides=overrides, device=dev, seed=0, auto_reset=True)
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    acts = torch.randint(0, 5, (8, E, env.N), generator=gen, device=dev, dtype=torch.int32)
    ptrs = [acts[i].data_ptr() for i in range(8)]
    env.reset()
    for i in range(40):
        env.step_raw(ptrs[i % 8])
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 100
    a.record()
    for i in range(K):
        env.step_raw(ptrs[i % 8])
    b.record()
    torch.cuda.synchronize(dev)
    ms = a.elapsed_time(b) / K
    gbs = ALGO_BYTES_PER_ENV_STEP * E / (ms * 1e-3) / 1e9
    out = {"envs": E, "kernel": "rg::tpe::step_kernel<PCP,N=5> (one lane per env)", "steps": K,
           "ms_per_step": ms, "agent_steps_per_s": E * env.N / (ms * 1e-3),
           "hbm_achieved_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}
    env.close()
    return out


def ipm_leg(dev, overrides, E=ENVS_PER_GPU, K=200):
    """Side measurement (not `value`): the headline workload with `barrier_solver: cvxopt` -- the certificate's QP evaluated as the
    reference's stack evaluates it (rps -> cvxopt's interior-point `qp` at reltol = feastol = 1e-2; utilities/controller.py:13-16,23),
    restated in binary64 (csrc/ipm_qp.h), instead of the exact projection the headline computes.  Same envs, same random policy,
    one rg_step launch per env step; ~10 interior-point iterations of a 10 x 10 KKT system per QP, two QPs per step."""
    import torch
    from marbler_amd import VecRobotariumEnv
    env = VecRobotariumEnv("PredatorCapturePrey", E, overrides=dict(overrides, barrier_solver="cvxopt"), device=dev, seed=0, auto_reset=True,
                           collect_qp_stats=True)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    acts = torch.randint(0, 5, (32, E, env.N), generator=gen, device=dev, dtype=torch.int32)
    ptrs = [acts[i].data_ptr() for i in range(32)]
    env.reset()
    for i in range(40):
        env.step_raw(ptrs[i % 32])
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(K):
        env.step_raw(ptrs[i % 32])
    b.record()
    torch.cuda.synchronize(dev)
    ms = a.elapsed_time(b) / K
    it = env.qp_sweeps.float()
    out = {"barrier_solver": "cvxopt (RG_QP_CVXOPT: restated interior-point iterate, binary64)", "envs": E, "steps": K,
           "kernel": "rg::step_kernel<PCP,GW=8,...,QPM=1> + rg::ipm::solve_qp<5,8>", "ms_per_step": ms,
           "agent_steps_per_s": E * env.N / (ms * 1e-3), "iterations_mean_of_step_max": float(it.mean()), "iterations_max": int(it.max()),
           "vs_exact_projection": "the headline (`value`) is the same workload with barrier_solver: exact"}
    env.close()
    return out


def actor_leg(dev, E=4096, N=4, D=16, H=128, A=5, bursts=7, reps=100):
    """Side measurement (not `value`): the fused policy-inference kernel of row f3 (csrc/actor_mfma.hip, rg_actor_forward) at
    the evaluation loop's shape -- PredatorCapturePrey's zoo actor: 4096 envs x 4 agents, 16 observation floats + agent id,
    GRU hidden 128, 5 actions, random weights -- with its own roofline.  The bound is the matrix cores, and `frac` prices the launch as
    what it issues.  Default form (round 5): float32 = two binary16 planes, THREE plane products per float32 product (robogym.h
    rg_actor_pack_gru_f16x2), counted as binary16 MFMA flops against the dense binary16 / bfloat16 peak.  The round-4 form (three
    bfloat16 planes, six products) is timed beside it: it issues twice the matrix-core work, so its `frac` is HIGHER and its launch
    LONGER -- the figure to compare across forms is `ms_per_launch`.  `f32_equivalent_*` is the network's own arithmetic
    (multiply-adds x 2 of fc1 + the two GRU matrices + fc2) against the float32 MFMA peak -- the side figure."""
    import torch
    from marbler_amd.evaluate import BatchedActor
    g = torch.Generator().manual_seed(3)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * 0.3  # noqa: E731
    I = D + N
    sd = {"fc1.weight": r(H, I), "fc1.bias": r(H), "rnn.weight_ih": r(3 * H, H), "rnn.weight_hh": r(3 * H, H), "rnn.bias_ih": r(3 * H),
          "rnn.bias_hh": r(3 * H), "fc2.weight": r(A, H), "fc2.bias": r(A)}
    obs = torch.rand(E, N, D, device=dev)
    q = torch.empty(E, N, A, device=dev)
    act = torch.empty(E, N, dtype=torch.int32, device=dev)
    flop = 2.0 * E * N * (I * H + 2 * 3 * H * H + H * A)
    forms = {}
    for form, products in (("f16x2", 3), ("bf16x3", 6)):
        actor = BatchedActor(sd, N, device=dev, pack_gru=form)
        hidden = torch.zeros(E, N, H, device=dev)
        for _ in range(20):
            actor.forward_fused(obs, hidden, q_out=q, actions_out=act)
        torch.cuda.synchronize(dev)
        times = []
        for _ in range(bursts):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                actor.forward_fused(obs, hidden, q_out=q, actions_out=act)
            b.record()
            torch.cuda.synchronize(dev)
            times.append(a.elapsed_time(b) / reps)
        ms = sorted(times)[len(times) // 2]
        issued = products * 2.0 * E * N * (2 * 3 * H * H) / (ms * 1e-3) / 1e12      # plane products per float32 product, as issued
        forms[form] = {"ms_per_launch": ms, "ms_per_launch_min": min(times), "plane_products_per_f32_product": products,
                       "issued_TFLOPs": issued, "frac_of_dense_16bit_mfma_peak": issued / BF16_MFMA_PEAK_TFLOPS,
                       "f32_equivalent_TFLOPs": flop / (ms * 1e-3) / 1e12}
    d = forms["f16x2"]
    return {"kernel": "rg::actor_kernel<128, 2> (rg_actor_forward, GRU on two binary16 planes)", "rows": E * N, "hidden": H, "ms_per_launch": d["ms_per_launch"],
            "ms_per_launch_min": d["ms_per_launch_min"], "flops_per_launch": flop,
            # priced as what the launch ISSUES: 16-bit MFMA work against the dense 16-bit peak; the network's own float32
            # arithmetic against the float32 MFMA peak is the side figure (it flatters: three cheap plane products per product)
            "roofline": {"bound": "mfma", "achieved": d["issued_TFLOPs"], "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": d["frac_of_dense_16bit_mfma_peak"],
                         "peak_kind": "dense binary16 MFMA (= the bfloat16 rate), as issued (the GRU's float32 products run as three binary16 plane "
                                      "products each; fc1 / fc2 are 28 float32 MFMAs per wave, not counted)",
                         "f32_equivalent_TFLOPs": d["f32_equivalent_TFLOPs"],
                         "f32_equivalent_frac_of_f32_mfma_peak": d["f32_equivalent_TFLOPs"] / F32_MFMA_PEAK_TFLOPS},
            "forms": forms, "agent_rows_per_s": E * N / (d["ms_per_launch"] * 1e-3)}


def launch_ranks(args):
    """`python bench.py --gpus N` without torchrun: start N ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment; one process per GPU, backend nccl unless --dist-backend says otherwise),
    relay rank 0's JSON line, exit non-zero if any rank fails or the rendezvous does not complete.
    FAIL FAST: all children are polled; the moment any rank exits non-zero the others are killed and that code is
    returned (a rank that dies before the rendezvous would otherwise leave rank 0 blocked in init_process_group until
    the time-out).  The time-out (RG_BENCH_LAUNCH_TIMEOUT, default 540 s) stays below the driver's own 600 s, so a
    stuck node reports an error of this script, not a killed command.
    The parent imports neither torch nor HIP: the children are fresh processes started before anything touches a GPU,
    nothing is exec'ed after a GPU has been initialised."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    argv = [a for a in sys.argv[1:]]
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")      # rank 0's stdout (a file: no pipe to drain while polling)
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + float(os.environ.get("RG_BENCH_LAUNCH_TIMEOUT", "540"))
    rc = 0
    try:
        while True:
            codes = [pr.poll() for pr in procs]
            failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if failed:
                for r, c in failed:
                    print(f"bench.py: rank {r} exited with {c}; stopping the other ranks", file=sys.stderr)
                rc = failed[0][1] if failed[0][1] > 0 else 1      # a signal (negative code) reports as 1
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                print("bench.py: ranks did not finish in time", file=sys.stderr)
                rc = 4
                break
            time.sleep(0.05)
    finally:
        for pr in procs:                        # exactly the processes started above
            if pr.poll() is None:
                pr.kill()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
    out0.seek(0)
    line = None
    for ln in out0.read().splitlines():
        if ln.startswith("{"):
            line = ln
    out0.close()
    if rc == 0 and line is None:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        rc = 5
    if rc == 0:
        print(line)
    return rc


def dry_run(args, rank, world, collective):
    """The N > 1 plumbing without a GPU: parameter block broadcast from rank 0, env sharding, the statistics gather
    and the max-over-ranks timing reduction, on whatever backend the ranks were started with (gloo on a CPU box)."""
    import torch
    import torch.distributed as dist
    from marbler_amd import load_config, make_params
    from marbler_amd import dist as rgdist
    from marbler_amd.params import params_to_bytes
    E = args.envs_per_gpu
    # only rank 0 holds the benchmark's overrides; the others build the scenario's default block and receive rank 0's
    params = make_params(args.scenario, load_config(args.scenario, overrides=bench_overrides(args.scenario) if rank == 0 else {}))
    if world > 1:
        params = rgdist.broadcast_params(params, src=0, device="cpu")
    offset, count = rgdist.shard(world * E, rank, world)
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed * 1e3]
    if world > 1:   # the same gather of every rank's own time as the measured run (per_rank_ms_per_step), then the max
        mine = torch.tensor([elapsed * 1e3], dtype=torch.float64)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        per_rank = [float(v[0]) for v in everyone]
        elapsed = max(per_rank) * 1e-3
    zeros_f, zeros_i = torch.zeros(count), torch.full((count,), rank, dtype=torch.int32)
    stats = rgdist.gather_episode_stats(zeros_f, zeros_i, zeros_i.clone(), dst=0, total_envs=world * E)
    if rank == 0:
        import hashlib
        out = {"metric": "env agent-steps/sec", "value": None, "unit": "agent-steps/s", "n_gpus": world, "steps": 0,
               "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "per_rank_ms_per_step": per_rank,   # (dry run: each rank's time in the barrier, to exercise the gather)
               "dtype": "f32", "data": "none (dry run: no GPU work)", "dry_run": True,
               "config": {"workload": f"{args.scenario}-v0, {E} envs x {params.n_agents} agents per GPU (dry run)",
                          "envs_per_gpu": E, "agents": int(params.n_agents), "parallelism": f"env-sharded x{world}"},
               "params_sha1": hashlib.sha1(params_to_bytes(params)).hexdigest(), "shard_of_rank_0": [offset, count],
               "gathered_envs": int(stats[0].numel()), "gathered_rank_ids": sorted(set(int(v) for v in stats[1]))}
        if collective is not None:
            out["collective"] = collective
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-saturated", action="store_true", help="skip the 524288-env side measurement")
    ap.add_argument("--no-graph", action="store_true",
                    help="skip the graph_replay side measurement (profiling runs: its launches would mix into the headline kernel's row)")
    ap.add_argument("--counters-child", action="store_true", help=argparse.SUPPRESS)   # a rocprofv3 --pmc pass of live_counters()
    ap.add_argument("--scenario", default="PredatorCapturePrey")
    ap.add_argument("--barrier-solver", default="exact", choices=("exact", "cvxopt"),
                    help="how the certificate's QP is evaluated (config key barrier_solver): exact = the projection (the headline); "
                         "cvxopt = the restated interior-point iterate the reference's stack computes (profiling runs of that mode)")
    ap.add_argument("--spinup-ms", type=float, default=1000.0,
                    help="untimed step launches before reset() + the W warm-up steps, so that a GPU that idled while the host did the CPU "
                         "baseline (or while the previous process exited) is at its clocks when the warm-up starts; 0 = none")
    ap.add_argument("--dist-backend", default=None, help="nccl (default, = RCCL) | gloo (CPU rehearsal of the N>1 path)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (a 1-GPU box); never for a measured run")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: ranks rendezvous, broadcast the parameter block, shard the envs, gather "
                         "(empty) statistics and print the line with value null -- the N > 1 plumbing on a CPU box")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))            # this process never touches a GPU
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: refusing to report a line "
              f"for a different number of ranks", file=sys.stderr)
        sys.exit(2)

    if os.environ.get("RG_BENCH_FAULT_RANK") == os.environ.get("RANK", "0") and "WORLD_SIZE" in os.environ:
        # fault injection for the launcher test (tests/test_host.py): this rank dies before the rendezvous
        print(f"bench.py: rank {os.environ.get('RANK')} told to fail (RG_BENCH_FAULT_RANK)", file=sys.stderr)
        sys.exit(7)

    # The CPU baseline runs first, while this process has not touched the GPU: it starts child
    # processes (one env each on the host cores), and nothing is exec'ed after HIP is initialised.
    if args.counters_child:
        return counters_child(args)
    cpu_ref = None
    live, live_src = {}, None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline and not args.dry_run:
        cpu_ref = cpu_baseline()
        if args.scenario == "PredatorCapturePrey" and args.envs_per_gpu == ENVS_PER_GPU and args.gpus == 1 and args.barrier_solver == "exact":
            live, live_src = live_counters(args)   # (also before this process touches the GPU)

    import torch
    import torch.distributed as dist
    from marbler_amd import VecRobotariumEnv, make_params, load_config
    from marbler_amd import dist as rgdist

    rank, world, local = rgdist.init_from_env(backend=args.dist_backend, device_index=0 if args.share_gpu else None)
    collective = None
    grouped = dist.is_initialized()     # world > 1, or one rank with RG_FORCE_PROCESS_GROUP=1 (the RCCL path on a one-GPU box)
    if grouped:
        # every rank reports in: an all_gather of the rank ids over the backend the run uses
        cdev = torch.device("cpu") if dist.get_backend() == "gloo" else torch.device("cuda", local)
        me = torch.tensor([rank], dtype=torch.int64, device=cdev)
        seen = [torch.zeros_like(me) for _ in range(world)]
        dist.all_gather(seen, me)
        ranks_seen = sorted(int(t.item()) for t in seen)
        collective = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen": ranks_seen,
                      "library": "RCCL over xGMI (torch.distributed nccl backend on ROCm)" if dist.get_backend() == "nccl"
                                 else "gloo over TCP loopback (rehearsal)"}
        if ranks_seen != list(range(args.gpus)):
            print(f"bench.py: ranks seen {ranks_seen}, expected 0..{args.gpus - 1}", file=sys.stderr)
            sys.exit(3)
    if args.dry_run:
        return dry_run(args, rank, world, collective)
    dev = torch.device("cuda", local if grouped else 0)
    torch.cuda.set_device(dev)
    E = args.envs_per_gpu
    K, W = args.steps, args.warmup

    overrides = bench_overrides(args.scenario)
    if args.barrier_solver != "exact":
        overrides["barrier_solver"] = args.barrier_solver
    # rank 0 reads the YAML; every rank gets the parameter block by RCCL broadcast
    params = make_params(args.scenario, load_config(args.scenario, overrides=overrides)) if rank == 0 else None
    if grouped:
        if rank != 0:
            params = make_params(args.scenario, load_config(args.scenario, overrides=overrides))  # shape only
        params = rgdist.broadcast_params(params, src=0, device=dev)
    env = VecRobotariumEnv(args.scenario, E, params=params, device=dev, seed=0, env_offset=rank * E,
                           auto_reset=True)
    N = env.N
    n_act = 20 if args.scenario == "MaterialTransport" else 5
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    n_batches = min(K + W, 512)   # distinct action batches, cycled (resident in HBM)
    actions = torch.randint(0, n_act, (n_batches, E, N), generator=gen, device=dev, dtype=torch.int32)
    ptrs = [actions[i].data_ptr() for i in range(n_batches)]
    env.reset()
    step = env.step_raw
    if args.spinup_ms > 0:   # not part of W or K: the same launches on scratch state, then the episode starts over
        t_spin = time.perf_counter()
        i = 0
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            for _ in range(64):
                step(ptrs[i % n_batches])
                i += 1
            torch.cuda.synchronize(dev)
        env.reset()

    def barrier():
        torch.cuda.synchronize(dev)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()                                # HIP events are created on first use: not inside the timed region
    ev1.record()
    for i in range(W):
        step(ptrs[i % n_batches])
    barrier()
    # The HIP events bracket exactly the K launches; the host clock brackets the K calls and the contract's synchronize.
    # ev0 is recorded on the idle stream just BEFORE the clock starts and the end is awaited by the synchronize alone:
    # measured (tools/fixed_probe2.py, 20 steps) a record inside the clocked region costs ~5 us of host time and polling
    # the end event ~10 us more than the plain synchronize -- 339 us for the region against 321-323 us this way, the
    # same as with no events at all.
    ev0.record()
    t0 = time.perf_counter()
    for i in range(K):
        rc = step(ptrs[(W + i) % n_batches])
    ev1.record()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0          # this rank's K steps, device-complete
    barrier()
    assert rc == 0
    gpu_ms_total = ev0.elapsed_time(ev1)
    # every rank's own clock and its own HIP-event time, so that a slow N-GPU line says WHICH rank was slow (the value is
    # still computed from the max over ranks, as the contract asks)
    per_rank = [[elapsed / K * 1e3, gpu_ms_total / K]]
    if grouped:
        mine = torch.tensor(per_rank[0], device=rgdist.collective_device(dev), dtype=torch.float64)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        per_rank = [[float(v[0]), float(v[1])] for v in everyone]
        elapsed = max(v[0] for v in per_rank) * K * 1e-3

    # ---- roofline leg: the same launches with a HIP event pair around each (stream = the launch stream)
    n_probe = min(K, 300)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_probe)]
    for i, (a, b) in enumerate(evs):
        a.record()
        step(ptrs[i % n_batches])
        b.record()
    torch.cuda.synchronize(dev)
    durs = sorted(a.elapsed_time(b) for a, b in evs)
    kernel_ms = sum(durs) / len(durs)
    kernel_ms_median = durs[len(durs) // 2]

    # episode statistics gathered to rank 0 (RCCL all_gather over xGMI when world > 1)
    stats = rgdist.gather_episode_stats(env.done_return_sum, env.done_count, env.done_steps_sum, dst=0, total_envs=world * E)

    if rank == 0:
        total_agent_steps = world * E * N * K
        value = total_agent_steps / elapsed
        bytes_per_launch = (ALGO_BYTES_PER_ENV_STEP if args.scenario == "PredatorCapturePrey"
                            else ALGO_BYTES_OTHER[args.scenario]) * E
        achieved = bytes_per_launch / (gpu_ms_total / K * 1e-3) / 1e9 if bytes_per_launch else None
        counters, counters_src = committed_counters() if (args.scenario == "PredatorCapturePrey" and E == ENVS_PER_GPU and args.barrier_solver == "exact") \
            else ({}, None)
        counters_kind = "from_committed_profile"
        if live:
            counters, counters_src, counters_kind = live, live_src, "measured_in_this_run"
        elif live_src:
            counters_src = f"{counters_src} (live passes: {live_src})"
        traffic = (FETCH_SIZE_SCALE * counters["FETCH_SIZE"] + WRITE_SIZE_SCALE * counters["WRITE_SIZE"]) * 1024.0 if counters else None
        kernel_s = gpu_ms_total / K * 1e-3
        valu = None
        if "SQ_INSTS_VALU" in counters:
            insts = counters["SQ_INSTS_VALU"]
            valu = {"insts_per_launch": insts, "waves_per_launch": counters.get("SQ_WAVES"),
                    "issue_util": insts * VALU_ISSUE_CYCLES / (kernel_s * CLOCK_HZ * SIMDS),
                    "formula": "SQ_INSTS_VALU x 4 issue cycles / (kernel time x 2.4 GHz x 1024 SIMDs)",
                    "kernel_time": "this run (HIP events)", "source": counters_src,
                    "note": "the binding roof: one wave per SIMD executes a dependent instruction chain "
                            "(~6 cycles per instruction measured, tools/ubench); launch time = the slowest wave"}
        out = {
            "metric": "env agent-steps/sec", "value": value, "unit": "agent-steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "spinup_ms": args.spinup_ms, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "per_rank_ms_per_step": [round(v[0], 6) for v in per_rank], "per_rank_kernel_ms_per_step": [round(v[1], 6) for v in per_rank],
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scenario}-v0, {E} envs x {N} agents per GPU, random policy, "
                                   f"auto-reset, update_frequency {env.params.update_frequency}, barrier_solver {args.barrier_solver}, "
                                   f"one rg_step launch per env step; "
                                   f"{args.spinup_ms:g} ms of untimed spin-up launches + reset() before the {W} warm-up steps",
                       "envs_per_gpu": E, "agents": N, "parallelism": f"env-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_kind": counters_kind + " (rocprofv3 --pmc passes, separate from the timed region): "
                                         "2 x FETCH_SIZE + WRITE_SIZE, the factors calibrated on known byte counts in the kernels' own "
                                         "access shapes (profiles/r3_hbm_calibration.csv); an upper bound: 0.17 MB reported (0.17-0.35 MB of "
                                         "bytes, factor uncalibrated for instruction fetches) is the kernel's own instructions, fetched by each "
                                         "of the 8 XCD L2s per launch (profiles/r3_fetch_floor_pmc_summary.csv)",
                         "traffic_source": counters_src, "valu": valu,
                         "kernel": (f"rg::step_kernel<{args.scenario},GW={4 if N <= 4 else 8 if N <= 8 else 16},N={N}> (lane group per env)"
                                    if env.step_kernel == "group"
                                    else f"rg::tpe::step_kernel<{args.scenario},N={N}> (one lane per env)"),
                         "kernel_ms_avg": gpu_ms_total / K,   # HIP events around the timed region / K (back-to-back launches)
                         "kernel_ms_avg_event_pair_per_launch": kernel_ms,
                         "kernel_ms_median_event_pair_per_launch": kernel_ms_median,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "latency/VALU-bound fused step (~120 flop/B): HBM fraction is structurally tiny, "
                                 "see DESIGN.md"},
        }
        if collective is not None:
            out["collective"] = collective
        if stats is not None:
            rs, cs, ss = stats
            n_ep = int(cs.sum().item())
            out["episodes"] = {"finished": n_ep,
                               "mean_return": float(rs.sum().item() / max(n_ep, 1)),
                               "mean_length": float(ss.sum().item() / max(n_ep, 1))}
        if world == 1 and not args.no_saturated and args.scenario == "PredatorCapturePrey" and args.barrier_solver == "exact":
            if not args.no_graph:
                out["graph_replay"] = graph_leg(env, dev, ptrs, min(K, 100))
            out["rollout"] = rollout_leg(env, dev, n_act, 777)
            out["host_boundary"] = host_boundary_leg(env, dev, n_act)
            out["saturated"] = saturated_leg(dev, overrides)
            out["actor"] = actor_leg(dev)
            out["interior_point_mode"] = ipm_leg(dev, overrides)
            out["saturated_2m"] = saturated_leg(dev, overrides, 4 * SATURATED_ENVS)
        if cpu_ref is not None:
            out["cpu_baseline"] = cpu_ref
        print(json.dumps(out))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
