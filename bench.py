#!/usr/bin/env python3
"""bench.py -- env agent-steps/sec of the HIP step engine (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

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


def measured_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r1_final_pmc_summary.csv: FETCH_SIZE and WRITE_SIZE in KB, separate --pmc runs).
    Counters cannot be read from inside an un-profiled run; returns (bytes, source) or (None, None)."""
    path = os.path.join(ROOT, "profiles", "r1_final_pmc_summary.csv")
    try:
        import csv
        kb = {}
        for r in csv.DictReader(open(path)):
            if r["counter"] in ("FETCH_SIZE", "WRITE_SIZE") and "rg::step_kernel<0, 8, false, 5, false>" in r["kernel"]:
                kb[r["counter"]] = float(r["mean_per_launch"])
        if len(kb) == 2:
            return (kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0, "profiles/r1_final_pmc_summary.csv"
    except OSError:
        pass
    return None, None


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


def cpu_all_cores(seconds=8.0, worker="_port_worker"):
    """the same port, one env per process, one process per host core (the shape of
    EPyMARL's parallel runner).  CPU-only child processes; returns (agent-steps/s, processes)."""
    import subprocess
    n = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))   # one GPU's share of the host: 16 cores
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
    # under rocprofv3 the profiler's preloaded library has already initialised the GPU in this process:
    # no child processes then (each would be an exec after GPU initialisation)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith("ROCPROF") for k in os.environ)
    all_rate, n_proc = (None, 0) if profiled else cpu_all_cores()
    c_all, _ = (None, 0) if profiled else cpu_all_cores(4.0, "_oracle_worker")
    return {"value": py_rate, "unit": "agent-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} env-steps of 1 env x 5 agents, NumPy float64 port in the reference's shape "
                      f"(oracle/np_port.py), {dt:.1f} s on one core",
            "cpu_model": cpu_model(), "host_cores": os.cpu_count(),
            "all_cores": {"value": all_rate, "cores": n_proc, "kind": "port",
                          "sample": f"{n_proc} processes x 1 env each (EPyMARL parallel-runner shape; one GPU's share of "
                                    f"the host cores), 8 s" if not profiled else "skipped under rocprofv3 (no child processes)"},
            "c_oracle_f64_1core": c_rate, "c_oracle_f64_all_cores": {"value": c_all, "cores": n_proc}}


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


def saturated_leg(dev, overrides):
    """The same step at a batch that fills the chip (524288 envs, thread-per-env kernel): where the
    path stands against the HBM roof when launch latency no longer binds.  Reported beside the
    headline, never as `value`."""
    import torch
    from marbler_amd import VecRobotariumEnv
    E = SATURATED_ENVS
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-saturated", action="store_true", help="skip the 524288-env side measurement")
    ap.add_argument("--scenario", default="PredatorCapturePrey")
    ap.add_argument("--dist-backend", default=None, help="nccl (default, = RCCL) | gloo (CPU rehearsal of the N>1 path)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (a 1-GPU box); never for a measured run")
    args = ap.parse_args()

    # The CPU baseline runs first, while this process has not touched the GPU: it starts child
    # processes (one env each on the host cores), and nothing is exec'ed after HIP is initialised.
    cpu_ref = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        cpu_ref = cpu_baseline()

    import torch
    import torch.distributed as dist
    from marbler_amd import VecRobotariumEnv, make_params, load_config
    from marbler_amd import dist as rgdist

    rank, world, local = rgdist.init_from_env(backend=args.dist_backend, device_index=0 if args.share_gpu else None)
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    E = args.envs_per_gpu
    K, W = args.steps, args.warmup

    overrides = PCP_OVERRIDES if args.scenario == "PredatorCapturePrey" else \
        {"n_agents": 8} if args.scenario == "Warehouse" else \
        {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}
    # rank 0 reads the YAML; every rank gets the parameter block by RCCL broadcast
    params = make_params(args.scenario, load_config(args.scenario, overrides=overrides)) if rank == 0 else None
    if world > 1:
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

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(W):
        step(ptrs[i % n_batches])
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(K):
        rc = step(ptrs[(W + i) % n_batches])
    ev1.record()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0          # this rank's K steps, device-complete
    barrier()
    assert rc == 0
    gpu_ms_total = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([elapsed], device=rgdist.collective_device(dev), dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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
    stats = rgdist.gather_episode_stats(env.done_return_sum, env.done_count, env.done_steps_sum, dst=0)

    if rank == 0:
        total_agent_steps = world * E * N * K
        value = total_agent_steps / elapsed
        bytes_per_launch = (ALGO_BYTES_PER_ENV_STEP if args.scenario == "PredatorCapturePrey"
                            else ALGO_BYTES_OTHER[args.scenario]) * E
        achieved = bytes_per_launch / (gpu_ms_total / K * 1e-3) / 1e9 if bytes_per_launch else None
        traffic, traffic_src = measured_traffic() if (args.scenario == "PredatorCapturePrey" and E == ENVS_PER_GPU) \
            else (None, None)
        out = {
            "metric": "env agent-steps/sec", "value": value, "unit": "agent-steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scenario}-v0, {E} envs x {N} agents per GPU, random policy, "
                                   f"auto-reset, update_frequency {env.params.update_frequency}, one rg_step launch per env step",
                       "envs_per_gpu": E, "agents": N, "parallelism": f"env-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": (f"rg::step_kernel<{args.scenario},GW={4 if N <= 4 else 8 if N <= 8 else 16},N={N}> (lane group per env)"
                                    if (args.scenario != "PredatorCapturePrey" or E < 40960)
                                    else "rg::tpe::step_kernel<PCP,N=5> (one lane per env)"),
                         "kernel_ms_avg": gpu_ms_total / K,   # HIP events around the timed region / K (back-to-back launches)
                         "kernel_ms_avg_event_pair_per_launch": kernel_ms,
                         "kernel_ms_median_event_pair_per_launch": kernel_ms_median,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "latency/VALU-bound fused step (~120 flop/B): HBM fraction is structurally tiny, "
                                 "see DESIGN.md"},
        }
        if stats is not None:
            rs, cs, ss = stats
            n_ep = int(cs.sum().item())
            out["episodes"] = {"finished": n_ep,
                               "mean_return": float(rs.sum().item() / max(n_ep, 1)),
                               "mean_length": float(ss.sum().item() / max(n_ep, 1))}
        if world == 1 and not args.no_saturated and args.scenario == "PredatorCapturePrey":
            out["rollout"] = rollout_leg(env, dev, n_act, 777)
            out["saturated"] = saturated_leg(dev, overrides)
        if cpu_ref is not None:
            out["cpu_baseline"] = cpu_ref
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
