#!/usr/bin/env python3
"""Divergence cost of the thread-per-env kernel (diagnostic build -DRG_TPE_DIAG, never shipped):
per 64-env wave, the sweeps the wave executes (max over its lanes) and the chunks it replays."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ROBOGYM_LIB"] = os.path.join(ROOT, "marbler_amd", "librobogym_diag.so")
os.environ["RG_STEP_KERNEL"] = "tpe"
sys.path.insert(0, ROOT)
import torch
from marbler_amd import VecRobotariumEnv
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, collect_qp_stats=True)
acts = torch.randint(0, 5, (64, E, 5), device=env.device, dtype=torch.int32)
env.reset()
tot = {"sw1_env": 0.0, "sw2_env": 0.0, "sw1_wave": 0.0, "sw2_wave": 0.0, "replay_env": 0.0, "replay_wave": 0.0}
n = 0
for i in range(150):
    env.step(acts[i % 64])
    if i >= 100:
        d = env.qp_sweeps.view(-1, 64)
        sw1, sw2, mask = (d >> 8) & 255, d & 255, d >> 16
        wave_mask = torch.zeros(d.shape[0], dtype=torch.int32, device=d.device)
        for b in range(6):
            wave_mask |= (((mask >> b) & 1).max(dim=1).values << b).int()
        pop = lambda m: sum(((m >> b) & 1) for b in range(6)).float()
        tot["sw1_env"] += float(sw1.float().mean()); tot["sw2_env"] += float(sw2.float().mean())
        tot["sw1_wave"] += float(sw1.max(dim=1).values.float().mean()); tot["sw2_wave"] += float(sw2.max(dim=1).values.float().mean())
        tot["replay_env"] += float(pop(mask).mean()); tot["replay_wave"] += float(pop(wave_mask).mean())
        n += 1
print({k: round(v / n, 3) for k, v in tot.items()})
d = env.qp_sweeps
for name, sw in (("QP1", (d >> 8) & 255), ("QP2", d & 255)):
    h = torch.bincount(sw.flatten(), minlength=20).float()
    h = h / h.sum()
    print(name, "sweep histogram (fraction of envs):", [round(float(x), 5) for x in h[:20]])
    for g in (8, 64):
        m = sw.view(-1, g).max(dim=1).values
        hh = torch.bincount(m, minlength=20).float()
        print(f"   max over {g} envs:", [round(float(x), 4) for x in (hh / hh.sum())[:20]])
