#!/usr/bin/env python3
"""Per-phase wave cycles inside a multi-step launch (rg_rollout), diagnostic -DRG_STAMPS build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ROBOGYM_LIB"] = os.path.join(ROOT, "marbler_amd", "librobogym_stamps.so")
os.environ["RG_STEP_KERNEL"] = "group"
sys.path.insert(0, ROOT)
import torch
from marbler_amd import VecRobotariumEnv
E, K = 4096, 64
env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, collect_qp_stats=True)
acts = torch.randint(0, 5, (K, E, 5), device=env.device, dtype=torch.int32)
env.reset()
buf = env.rollout(acts)
acc = torch.zeros(8, dtype=torch.float64)
n = 0
for rep in range(4):
    buf = env.rollout(acts, out=buf)
    s = buf["qp_sweeps"].view(K, -1, 8).double()
    acc += s[1:].mean(dim=(0, 1)).cpu()   # steps after the first of a launch
    first = s[0].mean(dim=0).cpu()
    n += 1
acc /= n
names = ["loaded", "ctrl1", "period1", "periods", "epilogue", "stored", "reset"]
prev = 0.0
for k in range(7):
    print(f"{names[k]:10s} cum {acc[k]:9.0f}  delta {acc[k]-prev:8.0f}   (first step of a launch: cum {first[k]:9.0f})")
    prev = acc[k]
