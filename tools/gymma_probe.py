#!/usr/bin/env python3
"""Cost of one GymmaVecEnv.step() (EPyMARL's gymma contract for E envs at once) against the bare env step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marbler_amd.gymma import GymmaVecEnv
for limit, alias in ((1000, False), (60, False), (60, True)):
    v = GymmaVecEnv("robotarium_gym:PredatorCapturePrey-v0", 4096, time_limit=limit, overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=1,
                    alias_outputs=alias)
    acts = torch.randint(0, 5, (64, 4096, 5), device=v.env.device, dtype=torch.int32)
    v.reset()
    for i in range(100):
        v.step(acts[i % 64]); v.get_obs(); v.get_state()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 1000
    for i in range(K):
        r, term, info = v.step(acts[i % 64])
        o = v.get_obs(); s = v.get_state()
    torch.cuda.synchronize()
    print(f"time_limit {limit}, alias_outputs {alias}: GymmaVecEnv.step + get_obs + get_state {1e6 * (time.perf_counter() - t0) / K:.1f} us per step of 4096 envs")
