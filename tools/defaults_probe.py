#!/usr/bin/env python3
"""The reference's DEFAULT scenario configurations (no overrides: what `gym.make('robotarium_gym:<S>-v0')` builds) at 4096 and
32768 envs: microseconds per rg_step launch and agent-steps/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marbler_amd import VecRobotariumEnv
for scn, nact in (("PredatorCapturePrey", 5), ("Warehouse", 5), ("MaterialTransport", 20), ("Simple", 5), ("ArcticTransport", 5)):
    for E in (4096, 32768):
        env = VecRobotariumEnv(scn, E, seed=1)
        acts = torch.randint(0, nact, (64, E, env.N), device=env.device, dtype=torch.int32)
        env.reset()
        for i in range(200):
            env.step_raw(acts[i % 64].data_ptr())
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(1000):
            env.step_raw(acts[i % 64].data_ptr())
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b)
        print(json.dumps({"scenario": scn, "N": env.N, "E": E, "kernel": env.step_kernel, "us_per_step": us,
                          "agent_steps_per_s": E * env.N / (us * 1e-6), "update_frequency": int(env.params.update_frequency)}), flush=True)
        env.close()
