#!/usr/bin/env python3
"""Where a wave spends its cycles: runs the -DRG_STAMPS diagnostic build (never the shipped
library) and prints per-phase wave-cycle shares (s_memtime ticks).  Build it first:
    python -c "from marbler_amd import build; build.build(defines=('RG_STAMPS',), out='marbler_amd/librobogym_stamps.so')"
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ROBOGYM_LIB"] = os.environ.get("RG_STAMPS_LIB") or os.path.join(ROOT, "marbler_amd", "librobogym_stamps.so")
sys.path.insert(0, ROOT)
import torch
from marbler_amd import VecRobotariumEnv
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SCN = sys.argv[2] if len(sys.argv) > 2 else "PredatorCapturePrey"   # PredatorCapturePrey | Warehouse | MaterialTransport
OV = {"PredatorCapturePrey": {"predator": 3, "capture": 2, "n_agents": 5}, "Warehouse": {"n_agents": 8},
      "MaterialTransport": {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}}[SCN]
TPE = os.environ.get("RG_STEP_KERNEL") == "tpe"   # thread-per-env kernel: 64 envs per wave (E a multiple of 64)
env = VecRobotariumEnv(SCN, E, overrides=OV, collect_qp_stats=True)   # (the stamps build runs full wavefronts: 8 envs per wave)
acts = torch.randint(0, 20 if SCN == "MaterialTransport" else 5, (64, E, env.N), device=env.device, dtype=torch.int32)
print(f"{SCN} {E} x {env.N}, {'thread-per-env' if TPE else 'lane-group'} kernel")
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
if os.environ.get("RG_STAMPS_CLOCK"):   # a -DRG_STAMPS_CLOCK build: slot 5 holds the wave's life in 100 MHz ticks
    print(f"shader clock while the waves ran: {float(acc[6] / acc[5]) * 100:.0f} MHz (mean wave life {float(acc[6]):.0f} cycles = {float(acc[5]) / 100:.2f} us)")
# the launch lasts as long as its slowest wave: distribution of the wave end stamps (slot 6) over the last launch
end = (env.qp_sweeps.view(-1, 64)[:, 6] if TPE else env.qp_sweeps.view(-1, 8)[:, 6]).double().sort(descending=True).values
print("last launch, wave end ticks: slowest %.0f, 4th %.0f, 16th %.0f, median %.0f, mean %.0f (%d waves)" %
      (end[0], end[min(3, len(end) - 1)], end[min(15, len(end) - 1)], end[len(end) // 2], end.mean(), len(end)))
