#!/usr/bin/env python3
"""Where the thread-per-env kernel overtakes the lane-group kernel, per agent count: us per step of both at a ladder
of batch sizes (robogym_capi.hip: tpe_min_envs).    python tools/crossover_probe.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marbler_amd import VecRobotariumEnv
CFG = {4: [("PredatorCapturePrey", {"predator": 2, "capture": 2, "n_agents": 4}, 5), ("MaterialTransport", {}, 20)],
       5: [("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5)],
       6: [("PredatorCapturePrey", {"predator": 3, "capture": 3, "n_agents": 6}, 5),
           ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20),
           ("Warehouse", {"n_agents": 6}, 5)],
       3: [("PredatorCapturePrey", {"predator": 2, "capture": 1, "n_agents": 3}, 5)]}
for N in [int(v) for v in sys.argv[1:]] or [6]:
    for scn, ov, nact in CFG[N]:
        for E in (16384, 24576, 32768, 49152, 65536, 98304, 131072, 196608):
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
