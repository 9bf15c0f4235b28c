#!/usr/bin/env python3
"""rg_rollout at the headline size, repeated: us per env step (64 steps per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marbler_amd import VecRobotariumEnv
E, K = 4096, 64
env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=0)
g = torch.Generator(device=env.device); g.manual_seed(777)
acts = torch.randint(0, 5, (K, E, 5), generator=g, device=env.device, dtype=torch.int32)
env.reset()
buf = env.rollout(acts)
for rep in range(6):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(16):
        env.rollout(acts, out=buf)
    b.record()
    torch.cuda.synchronize()
    print(f"{os.environ.get('ROBOGYM_LIB', 'librobogym_hip.so').split('/')[-1]}: rollout {a.elapsed_time(b) / (16 * K) * 1e3:.3f} us/step", flush=True)
# and single steps
ptrs = [acts[i].data_ptr() for i in range(K)]
for rep in range(3):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(2000):
        env.step_raw(ptrs[i % K])
    b.record()
    torch.cuda.synchronize()
    print(f"   rg_step {a.elapsed_time(b) / 2000 * 1e3:.3f} us/step", flush=True)
