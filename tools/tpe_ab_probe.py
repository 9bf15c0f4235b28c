#!/usr/bin/env python3
"""Thread-per-env kernel timing for A/B runs of library builds (ROBOGYM_LIB): us per step at a few chip-filling batch sizes
for N = 4, 5, 6.    python tools/tpe_ab_probe.py [N ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RG_STEP_KERNEL"] = "tpe"
import torch
from marbler_amd import VecRobotariumEnv
CFG = {4: [("PredatorCapturePrey", {"predator": 2, "capture": 2, "n_agents": 4}, 5), ("MaterialTransport", {}, 20)],
       5: [("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5)],
       7: [("Warehouse", {"n_agents": 7}, 5)],
       8: [("Warehouse", {"n_agents": 8}, 5)],
       6: [("PredatorCapturePrey", {"predator": 3, "capture": 3, "n_agents": 6}, 5),
           ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20),
           ("Warehouse", {"n_agents": 6}, 5)]}
for N in [int(v) for v in sys.argv[1:]] or [5, 6]:
    for scn, ov, nact in CFG[N]:
        for E in (131072, 262144, 524288):
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
            print(json.dumps({"lib": os.environ.get("ROBOGYM_LIB", "shipped").split("/")[-1], "scenario": scn, "N": N, "E": E,
                              "us_per_step": a.elapsed_time(b) / 60 * 1e3, "G_agent_steps_per_s": E * N / (a.elapsed_time(b) / 60 * 1e-3) / 1e9}), flush=True)
            env.close()
