#!/usr/bin/env python3
"""Which wavefront finishes last, and why: per launch, the stamps of the slowest wave
(diagnostic -DRG_STAMPS build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ROBOGYM_LIB"] = os.path.join(ROOT, "marbler_amd", "librobogym_stamps.so")
sys.path.insert(0, ROOT)
import torch
from marbler_amd import VecRobotariumEnv
E = 4096
env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, collect_qp_stats=True)
acts = torch.randint(0, 5, (64, E, 5), device=env.device, dtype=torch.int32)
env.reset()
names = ["loaded", "ctrl1", "period1", "periods", "epilogue", "stored", "reset", "sweeps"]
rows = []
for i in range(400):
    env.step(acts[i % 64])
    if i >= 100:
        s = env.qp_sweeps.view(-1, 8).double().cpu()
        w = int(s[:, 6].argmax())
        rows.append(torch.cat([s[w], s.mean(0)[6:7]]))
r = torch.stack(rows)
m = r.mean(0)
print("slowest wave per launch, mean over launches (cumulative ticks):")
prev = 0
for k in range(7):
    print(f"  {names[k]:9s} {m[k]:8.0f}  delta {m[k]-prev:8.0f}")
    prev = m[k]
print("  its max_sweeps (wave-leading env only):", float(m[7]), " | mean wave end:", float(m[8]))
d = r[:, 1:7] - r[:, 0:6]
print("share of slowest-wave time by phase:", [round(float(x), 3) for x in (torch.cat([r[:, :1], d], 1).mean(0) / m[6])])
