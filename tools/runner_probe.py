#!/usr/bin/env python3
"""BatchedRunner.run (gymma data collection for E envs: policy step, env step, the transition tensors EPyMARL stores): us per time step.
    python tools/runner_probe.py [envs=4096] [T=200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from marbler_amd.evaluate import BatchedActor
from marbler_amd.gymma import BatchedRunner, GymmaVecEnv
from test_gpu_actor import _random_actor
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for eps in (0.0, 0.1):
    for H in (64, 128):
        v = GymmaVecEnv("robotarium_gym:PredatorCapturePrey-v0", E, time_limit=1000, seed=3)
        actor = BatchedActor(_random_actor(1, v.obs_size + v.n_agents, H, v.n_actions, True, seed=2), v.n_agents, device=v.env.device)
        runner = BatchedRunner(v, actor, epsilon=eps, seed=1)
        runner.run(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b = runner.run(T)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / T * 1e6
        print(f"E {E} hidden {H} epsilon {eps}: {us:.1f} us per time step, {E * v.n_agents / us:.1f} M agent-steps/s collected, reward sum {float(b['reward'].sum()):.3f}", flush=True)
        v.env.close()
