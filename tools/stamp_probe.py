#!/usr/bin/env python3
"""Where a wave spends its cycles: runs the -DRG_STAMPS diagnostic build (never the shipped
library) and prints per-phase wave-cycle shares (s_memtime ticks).  Build it first:
    python -c "from marbler_amd import build; build.build(defines=('RG_STAMPS',), out='marbler_amd/librobogym_stamps.so')"
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ROBOGYM_LIB"] = os.path.join(ROOT, "marbler_amd", "librobogym_stamps.so")
sys.path.insert(0, ROOT)
import torch
from marbler_amd import VecRobotariumEnv
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
TPE = os.environ.get("RG_STEP_KERNEL") == "tpe"   # thread-per-env kernel: 64 envs per wave (E a multiple of 64)
env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, collect_qp_stats=True)
acts = torch.randint(0, 5, (64, E, 5), device=env.device, dtype=torch.int32)
env.reset()
names = ["loaded", "ctrl1", "period1", "periods", "epilogue", "stored", "reset", "sweeps"]
acc = torch.zeros(8, dtype=torch.float64)
mx = torch.zeros(8, dtype=torch.float64)
n = 0
for i in range(300):
    env.step(acts[i % 64])
    if i >= 100:
        s = (env.qp_sweeps.view(-1, 64)[:, :8] if TPE else env.qp_sweeps.view(-1, 8)).double().cpu()
        acc += s.mean(0)
        mx = torch.maximum(mx, s.max(0).values)
        n += 1
acc /= n
prev = 0.0
for k in range(7):
    print(f"{names[k]:10s} cum {acc[k]:9.0f} ticks  delta {acc[k]-prev:9.0f}   (max cum over waves {mx[k]:9.0f})")
    prev = acc[k]
print("mean max_sweeps of wave-leading env", float(acc[7]), "max", float(mx[7]))
