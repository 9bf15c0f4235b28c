#!/usr/bin/env python3
"""Anatomy of a 20-launch timed region (the driver's --steps 20): wall-clock time stamps of each host action and the HIP
events' view, variants of where the events are recorded."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marbler_amd import VecRobotariumEnv
E, K = 4096, 20
env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=0)
acts = torch.randint(0, 5, (64, E, 5), device=env.device, dtype=torch.int32)
ptrs = [acts[i].data_ptr() for i in range(64)]
for variant in ("events_inside", "ev0_before_t0", "ev0_before_t0_sync_only", "events_inside_sync_only", "no_events"):
    best = None
    for rep in range(8):
        env.reset()
        for i in range(5):
            env.step_raw(ptrs[i])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        torch.cuda.synchronize()
        if variant.startswith("ev0_before_t0"):
            e0.record()
        t0 = time.perf_counter()
        if variant.startswith("events_inside"):
            e0.record()
        t_a = time.perf_counter()
        env.step_raw(ptrs[5])
        t_b = time.perf_counter()
        for i in range(1, K):
            env.step_raw(ptrs[(5 + i) % 64])
        t_c = time.perf_counter()
        if variant != "no_events":
            e1.record()
            if not variant.endswith("sync_only"):
                while not e1.query():
                    pass
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ev = e0.elapsed_time(e1) * 1e3 if variant != "no_events" else float("nan")
        r = ((t1 - t0) * 1e6, ev, (t_a - t0) * 1e6, (t_b - t_a) * 1e6, (t_c - t_b) * 1e6, (t1 - t_c) * 1e6)
        best = r if best is None or r[0] < best[0] else best
    print(f"{variant:24s}: wall {best[0]:7.1f} us ({best[0] / K:5.2f} per step); events {best[1]:7.1f} us; ev0.record {best[2]:5.1f}, first launch call {best[3]:5.1f}, "
          f"other 19 calls {best[4]:6.1f}, wait for the end {best[5]:6.1f}")
