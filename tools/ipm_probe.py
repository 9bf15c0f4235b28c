#!/usr/bin/env python3
"""The step launch in both barrier-QP modes (config key `barrier_solver`: exact = the projection, cvxopt = the restated interior-point
iterate of csrc/ipm_qp.h): microseconds per rg_step launch at the BASELINE shapes, agent-steps per second, mean / max iterations.

    python tools/ipm_probe.py [--steps 200] > gpurun_out/r5/ipm_probe.jsonl
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marbler_amd import VecRobotariumEnv  # noqa: E402

SHAPES = [("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 4096, 5),
          ("PredatorCapturePrey", {}, 4096, 5),
          ("Warehouse", {"n_agents": 8}, 4096, 5),
          ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 2048, 20),
          ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 32768, 5),
          ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 512, 5)]

def cross(steps=30):
    """Both step kernels in the interior-point mode over a ladder of batch sizes (RG_STEP_KERNEL is read by rg_create): where the
    one-lane-per-env kernel takes over (robogym_capi.hip tpe_min_envs)."""
    dev = torch.device("cuda", 0)
    for scenario, ov, n_act in (("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5), ("PredatorCapturePrey", {}, 5)):
        for E in (4096, 8192, 16384, 24576, 32768, 65536, 131072):
            row = {"scenario": scenario, "agents": None, "envs": E}
            for kernel in ("group", "tpe"):
                os.environ["RG_STEP_KERNEL"] = kernel
                env = VecRobotariumEnv(scenario, E, overrides=dict(ov, barrier_solver="cvxopt"), device=dev, seed=0, auto_reset=True)
                row["agents"] = env.N
                assert env.step_kernel == kernel
                g = torch.Generator(device=dev).manual_seed(1234)
                acts = torch.randint(0, n_act, (8, E, env.N), generator=g, device=dev, dtype=torch.int32)
                env.reset()
                for i in range(10):
                    env.step_raw(acts[i % 8].data_ptr())
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(steps):
                    env.step_raw(acts[i % 8].data_ptr())
                b.record()
                torch.cuda.synchronize()
                row[kernel + "_us_per_step"] = round(a.elapsed_time(b) / steps * 1e3, 1)
                env.close()
            del os.environ["RG_STEP_KERNEL"]
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--cross", action="store_true", help="both kernels over a ladder of batch sizes instead of the BASELINE shapes")
    args = ap.parse_args()
    if args.cross:
        cross()
        sys.exit(0)
    dev = torch.device("cuda", 0)
    for scenario, ov, E, n_act in SHAPES:
        for solver in ("exact", "cvxopt"):
            env = VecRobotariumEnv(scenario, E, overrides=dict(ov, barrier_solver=solver), device=dev, seed=0, auto_reset=True, collect_qp_stats=True)
            g = torch.Generator(device=dev).manual_seed(1234)
            acts = torch.randint(0, n_act, (32, E, env.N), generator=g, device=dev, dtype=torch.int32)
            env.reset()
            for i in range(40):
                env.step_raw(acts[i % 32].data_ptr())
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(args.steps):
                env.step_raw(acts[i % 32].data_ptr())
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / args.steps * 1e3
            sw = env.qp_sweeps.float()
            print(json.dumps({"scenario": scenario, "envs": E, "agents": env.N, "barrier_solver": solver, "kernel": env.step_kernel,
                              "us_per_step": round(us, 2), "agent_steps_per_s": round(E * env.N / (us * 1e-6)),
                              "qp_iterations_or_sweeps_mean_of_step_max": round(float(sw.mean()), 2), "max": int(sw.max())}), flush=True)
            env.close()
