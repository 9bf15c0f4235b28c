#!/usr/bin/env python3
"""Distribution of wave end times within a launch (diagnostic -DRG_STAMPS build)."""
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
qs = []
for i in range(400):
    env.step(acts[i % 64])
    if i >= 100:
        s = env.qp_sweeps.view(-1, 8).double()
        end = s[:, 6].sort(descending=True).values
        qs.append(torch.stack([end[0], end[1], end[3], end[7], end[15], end[31], end[63], end[127], end[255], end[511], s[:, 6].mean()]).cpu())
q = torch.stack(qs).mean(0)
print("mean over launches of the k-th slowest wave's end tick:")
for name, v in zip(["top1", "top2", "top4", "top8", "top16", "top32", "top64", "top128", "median", "fastest", "mean"], q):
    print(f"  {name:8s} {float(v):8.0f}")
# phase breakdown of the slowest wave of each launch, and of the 8th slowest
names = ["loaded", "ctrl1", "period1", "periods", "epilogue", "stored", "reset"]
env.reset()
top, top8, cnt = torch.zeros(8, dtype=torch.float64), torch.zeros(8, dtype=torch.float64), 0
for i in range(400):
    env.step(acts[i % 64])
    if i >= 100:
        s = env.qp_sweeps.view(-1, 8).double()
        order = s[:, 6].argsort(descending=True)
        top += s[order[0]].cpu()
        top8 += s[order[7]].cpu()
        cnt += 1
for label, t in (("slowest wave", top / cnt), ("8th slowest", top8 / cnt)):
    prev = 0.0
    print(label + ": " + "  ".join(f"{n} +{float(t[k]) - (float(t[k-1]) if k else 0):.0f}" for k, n in enumerate(names)) + f"  (sweeps of its first env {float(t[7]):.1f})")
