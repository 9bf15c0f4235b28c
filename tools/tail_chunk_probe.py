#!/usr/bin/env python3
"""What the slowest wavefronts of a launch spend in the collision pre-test's fall-backs (diagnostic build
-DRG_STAMPS -DRG_STAMPS_CHUNK: slot 0 = ticks in the dense pre-test, 1 = ticks in the exact replay, 2 = dense chunks x 1000 +
replayed chunks, 3.. = the usual phase stamps).  RG_STAMPS_LIB names the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ROBOGYM_LIB"] = os.environ.get("RG_STAMPS_LIB") or os.path.join(ROOT, "marbler_amd", "librobogym_stamps_chunk.so")
sys.path.insert(0, ROOT)
import torch
from marbler_amd import VecRobotariumEnv
E = 4096
env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, collect_qp_stats=True)
acts = torch.randint(0, 5, (64, E, 5), device=env.device, dtype=torch.int32)
env.reset()
rows = []
for i in range(500):
    env.step(acts[i % 64])
    if i >= 100:
        s = env.qp_sweeps.view(-1, 8).double()
        order = s[:, 6].argsort(descending=True)
        rows.append(torch.cat([s[order[:8]].mean(0), s[order[len(order) // 2 - 4:len(order) // 2 + 4]].mean(0), s.mean(0)]).cpu())
m = torch.stack(rows).mean(0).view(3, 8)
for label, r in zip(("8 slowest waves of a launch", "8 median waves", "all waves"), m):
    print(f"{label:28s}: end {r[6]:7.0f} ticks; dense pre-test {r[0]:6.0f} ticks in {r[2] // 1000:4.2f} chunks; replay {r[1]:6.0f} ticks "
          f"in {r[2] % 1000:4.2f} chunks; periods done at {r[3]:7.0f}, epilogue {r[4] - r[3]:6.0f}, reset {r[6] - r[5]:5.0f}; sweeps of first env {r[7]:.1f}")
