#!/usr/bin/env python3
"""Where a wave of the fused actor kernel spends its time: -DRG_ACTOR_STAMPS build (phase stamps from s_memtime written over q).
    python tools/actor_stamps.py          # builds marbler_amd/librobogym_actor_stamps.so if needed (hipcc), then runs on cuda:0"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
LIB = os.path.join(ROOT, "marbler_amd", "librobogym_actor_stamps.so")
if "--build" in sys.argv or not os.path.exists(LIB):
    from marbler_amd import build
    build.build(defines=("RG_ACTOR_STAMPS",), out=LIB)
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["ROBOGYM_LIB"] = LIB
import numpy as np
import torch
from marbler_amd.evaluate import BatchedActor
from test_gpu_actor import _random_actor
N, D = 4, 16
for E, H in ((4096, 128), (4096, 64), (1024, 128)):
    actor = BatchedActor(_random_actor(1, D + N, H, 5, True, 3), N, device="cuda:0")
    obs = torch.rand(E, N, D, device="cuda:0")
    hidden = torch.zeros(E, N, H, device="cuda:0")
    q = torch.zeros(E, N, 5, device="cuda:0")
    for _ in range(5):
        actor.forward_fused(obs, hidden, q_out=q)
    torch.cuda.synchronize()
    waves = (E * N // 32) * (H // 32)
    st = q.view(torch.int32).flatten()[:waves * 8].cpu().numpy().reshape(waves, 8)[:, :7].astype(np.float64)
    names = ["staged", "fc1", "gru mfma", "gates", "hidden stored", "fc2", "argmax/q"]
    d = np.diff(np.concatenate([np.zeros((waves, 1)), st], axis=1), axis=1)
    print(f"E {E} H {H}: {waves} waves; s_memtime ticks (100 MHz) per phase, mean over waves [mean of wave 0 of each tile]:")
    for i, n in enumerate(names):
        print(f"   {n:14s} {d[:, i].mean():8.1f}   [{d[::H // 32, i].mean():8.1f}]   cumulative {st[:, i].mean():8.1f}")
