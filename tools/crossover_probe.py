#!/usr/bin/env python3
"""Where the thread-per-env kernel overtakes the lane-group kernel, per agent count: us per step of both at a ladder
of batch sizes (robogym_capi.hip: tpe_min_envs).    python tools/crossover_probe.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marbler_amd import VecRobotariumEnv
CFG = {2: [("PredatorCapturePrey", {"predator": 1, "capture": 1, "n_agents": 2, "num_neighbors": 1}, 5), ("Simple", {"n_agents": 2}, 5),
           ("Warehouse", {"n_agents": 2, "num_neighbors": 1}, 5)],
       3: [("PredatorCapturePrey", {"predator": 2, "capture": 1, "n_agents": 3}, 5), ("Warehouse", {"n_agents": 3}, 5)],
       4: [("PredatorCapturePrey", {"predator": 2, "capture": 2, "n_agents": 4}, 5), ("MaterialTransport", {}, 20),
           ("Simple", {}, 5), ("ArcticTransport", {}, 5), ("Warehouse", {"n_agents": 4}, 5)],
       5: [("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5), ("Warehouse", {"n_agents": 5}, 5),
           ("MaterialTransport", {"n_agents": 5, "n_fast_agents": 3, "n_slow_agents": 2, "start_dist": 0.25}, 20), ("Simple", {"n_agents": 5}, 5)],
       6: [("PredatorCapturePrey", {"predator": 3, "capture": 3, "n_agents": 6}, 5),
           ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20),
           ("Warehouse", {"n_agents": 6}, 5), ("Simple", {"n_agents": 6}, 5)]}
ONLY = os.environ.get("RG_CROSS_ONLY")   # e.g. "Simple,ArcticTransport": restrict to these scenarios
for N in [int(v) for v in sys.argv[1:]] or [6]:
    for scn, ov, nact in CFG[N]:
        if ONLY and scn not in ONLY.split(","):
            continue
        for E in (32768, 49152, 65536, 81920, 98304, 131072, 163840, 196608, 262144, 393216):
            row = []
            for kern in ("group", "tpe"):
                os.environ["RG_STEP_KERNEL"] = kern
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
                row.append(a.elapsed_time(b) / 60 * 1e3)
                env.close()
            print(f"{scn:20s} N={N} E={E:7d}  group {row[0]:8.2f} us   tpe {row[1]:8.2f} us   {'tpe' if row[1] < row[0] else 'group'}", flush=True)
