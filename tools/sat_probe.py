#!/usr/bin/env python3
"""The thread-per-env kernel at a chip-filling batch (PredatorCapturePrey, 5 agents): us per step and agent-steps/s.
    [ROBOGYM_LIB=<variant .so>] python tools/sat_probe.py [envs ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RG_STEP_KERNEL", "tpe")
import torch
from marbler_amd import VecRobotariumEnv
for E in [int(v) for v in sys.argv[1:]] or [524288]:
    env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=0)
    acts = torch.randint(0, 5, (8, E, 5), device=env.device, dtype=torch.int32)
    ptrs = [acts[i].data_ptr() for i in range(8)]
    env.reset()
    for i in range(40):
        env.step_raw(ptrs[i % 8])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(100):
        env.step_raw(ptrs[i % 8])
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 10
    print(f"{os.environ.get('ROBOGYM_LIB', 'librobogym_hip.so').split('/')[-1]:24s} E={E:8d}  {us:8.2f} us/step  {E * 5 / us * 1e-3:7.3f} G agent-steps/s", flush=True)
    env.close()
