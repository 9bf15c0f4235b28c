#!/usr/bin/env python3
"""Host-side launch cost vs device time (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marbler_amd import VecRobotariumEnv
for E in (8, 4096):
    env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5})
    acts = torch.randint(0, 5, (E, 5), device=env.device, dtype=torch.int32)
    env.reset()
    ptr = acts.data_ptr()
    for _ in range(200):
        env.step_raw(ptr)
    torch.cuda.synchronize()
    K = 2000
    t0 = time.perf_counter()
    for _ in range(K):
        env.step_raw(ptr)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"E={E}: host enqueue {1e6*(t1-t0)/K:.2f} us/call, total {1e6*(t2-t0)/K:.2f} us/step")
    # trivial torch kernel for comparison
    x = torch.zeros(64, device=env.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        x.add_(1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"   torch add_: host {1e6*(t1-t0)/K:.2f} us/call, total {1e6*(t2-t0)/K:.2f} us/step")
