#!/usr/bin/env python3
"""Fixed cost of a short timed region (the driver runs bench.py --steps 20 --warmup 5): wall clock around K launches
+ synchronize against the HIP-event time of the same launches, for several K and ways of waiting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marbler_amd import VecRobotariumEnv
E = 4096
env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=0)
acts = torch.randint(0, 5, (64, E, 5), device=env.device, dtype=torch.int32)
ptrs = [acts[i].data_ptr() for i in range(64)]
env.reset()
for i in range(5):
    env.step_raw(ptrs[i])
torch.cuda.synchronize()
for how in ("device_sync", "event_sync", "event_query_spin"):
    for K in (20, 200, 2000):
        best = None
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for i in range(K):
                env.step_raw(ptrs[i % 64])
            e1.record()
            t_enq = time.perf_counter()
            if how == "device_sync":
                torch.cuda.synchronize()
            elif how == "event_sync":
                e1.synchronize()
            else:
                while not e1.query():
                    pass
            t1 = time.perf_counter()
            r = ((t1 - t0) / K * 1e6, e0.elapsed_time(e1) / K * 1e3, (t_enq - t0) / K * 1e6)
            best = r if best is None or r[0] < best[0] else best
        print(f"{how:18s} K={K:5d}  wall {best[0]:7.2f} us/step   events {best[1]:7.2f} us/step   enqueue {best[2]:6.2f} us/step   fixed ~{(best[0]-best[1])*K:7.1f} us")
