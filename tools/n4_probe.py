#!/usr/bin/env python3
"""Thread-per-env kernel at N <= 4 and chip-filling batches (three waves per SIMD asked for): us per step.
    [ROBOGYM_LIB=<variant .so>] python tools/n4_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RG_STEP_KERNEL", "tpe")
import torch
from marbler_amd import VecRobotariumEnv
CASES = [("PredatorCapturePrey", {"predator": 2, "capture": 2, "n_agents": 4}, 5),
         ("MaterialTransport", {}, 20), ("Simple", {}, 5), ("ArcticTransport", {}, 5),
         ("PredatorCapturePrey", {"predator": 2, "capture": 1, "n_agents": 3}, 5)]
if "--n6" in sys.argv:
    CASES = [("PredatorCapturePrey", {"predator": 3, "capture": 3, "n_agents": 6}, 5),
             ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20),
             ("Warehouse", {"n_agents": 6}, 5)]
for scn, ov, nact in CASES:
    for E in (131072, 524288):
        env = VecRobotariumEnv(scn, E, overrides=ov, seed=0)
        acts = torch.randint(0, nact, (8, E, env.N), device=env.device, dtype=torch.int32)
        ptrs = [acts[i].data_ptr() for i in range(8)]
        env.reset()
        for i in range(30):
            env.step_raw(ptrs[i % 8])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(60):
            env.step_raw(ptrs[i % 8])
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 60 * 1e3
        print(f"{os.environ.get('ROBOGYM_LIB', 'librobogym_hip.so').split('/')[-1]:22s} {scn:20s} N={env.N} E={E:7d} {us:8.2f} us/step {E * env.N / us * 1e-3:7.3f} G", flush=True)
        env.close()
