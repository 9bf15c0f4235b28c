#!/usr/bin/env python3
"""The fused actor kernel alone (for rocprofv3): 200 launches at 4096 envs x 4 agents, hidden 128 and 64, in the weight forms
named on the command line (default: the default form), with the largest deviation from a float64 evaluation of the same network.
    python tools/actor_probe.py [f16x2] [bf16x3] [f32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from marbler_amd.evaluate import BatchedActor, DEFAULT_PACK_GRU
from test_gpu_actor import _random_actor
E, N, D = 4096, 4, 16
forms = [a for a in sys.argv[1:] if a in ("f16x2", "bf16x3", "f32")] or [DEFAULT_PACK_GRU]


def f64_reference(sd, obs, hidden, N):
    """rnn_agent.py:21-29 in float64 on the device (shared weights)."""
    w = {k: v.double().cuda() for k, v in sd.items()}
    eye = torch.eye(N, device="cuda:0", dtype=torch.float64).unsqueeze(0).expand(obs.shape[0], N, N)
    x = torch.relu(torch.cat([obs.double(), eye], dim=2) @ w["fc1.weight"].T + w["fc1.bias"])
    gi, gh = x @ w["rnn.weight_ih"].T + w["rnn.bias_ih"], hidden.double() @ w["rnn.weight_hh"].T + w["rnn.bias_hh"]
    H = hidden.shape[2]
    r, z = torch.sigmoid(gi[..., :H] + gh[..., :H]), torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    h = (1 - z) * n + z * hidden.double()
    return h @ w["fc2.weight"].T + w["fc2.bias"], h


for form in forms:
    for H in (128, 64):
        sd = _random_actor(1, D + N, H, 5, True, 3)
        actor = BatchedActor(sd, N, device="cuda:0", pack_gru=form)
        g = torch.Generator(device="cuda:0").manual_seed(5)
        obs = torch.rand(E, N, D, device="cuda:0", generator=g) * 3 - 1.5
        hidden = torch.rand(E, N, H, device="cuda:0", generator=g) * 2 - 1
        q64, h64 = f64_reference(sd, obs, hidden, N)
        h_t = hidden.clone()
        eye = torch.eye(N, device="cuda:0").unsqueeze(0).expand(E, N, N)
        q_t, h_t = actor.forward(torch.cat([obs, eye], dim=2), h_t)
        q, _ = actor.forward_fused(obs, hidden)
        torch.cuda.synchronize()
        err = (float((q.double() - q64).abs().max()), float((hidden.double() - h64).abs().max()))
        err_t = (float((q_t.double() - q64).abs().max()), float((h_t.double() - h64).abs().max()))
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
        print(f"{form} hidden {H}: {us:.1f} us per launch, {flop / us / 1e6:.1f} TFLOP/s of the network's float32 arithmetic; "
              f"max |q - f64| {err[0]:.2e}, |h - f64| {err[1]:.2e} (torch float32: {err_t[0]:.2e}, {err_t[1]:.2e})", flush=True)
