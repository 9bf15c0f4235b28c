#!/usr/bin/env python3
"""The fused actor kernel alone (for rocprofv3): 200 launches at 4096 envs x 4 agents, hidden 128 and 64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from marbler_amd.evaluate import BatchedActor
from test_gpu_actor import _random_actor
E, N, D = 4096, 4, 16
for H in (128, 64):
    actor = BatchedActor(_random_actor(1, D + N, H, 5, True, 3), N, device="cuda:0")
    obs = torch.rand(E, N, D, device="cuda:0")
    hidden = torch.zeros(E, N, H, device="cuda:0")
    for _ in range(20):
        actor.forward_fused(obs, hidden)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        actor.forward_fused(obs, hidden)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 200
    flop = 2.0 * E * N * ((D + N) * H + 2 * 3 * H * H + H * 5)
    print(f"hidden {H}: {us:.1f} us per launch, {flop / us / 1e6:.1f} TFLOP/s (f32 MFMA peak 157)", flush=True)
