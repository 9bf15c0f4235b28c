#!/usr/bin/env python3
"""Where a wave of the fused actor kernel spends its time: a -DRG_ACTOR_STAMPS build of csrc/actor_mfma.hip alone (phase stamps
from s_memtime written behind q; python tools/actor_lab/lab.py build v2s="-DRG_ACTOR_STAMPS").
    python tools/actor_stamps.py [--lib tools/actor_lab/build/actor_v2s.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
LIB = os.path.join(ROOT, "tools", "actor_lab", "build", "actor_v2s.so")   # python tools/actor_lab/lab.py build v2s="-DRG_ACTOR_STAMPS"
if "--lib" in sys.argv:
    LIB = sys.argv[sys.argv.index("--lib") + 1]
import numpy as np
import torch
from marbler_amd.evaluate import BatchedActor
from test_gpu_actor import _random_actor
import ctypes
from marbler_amd import _lib
lib = ctypes.CDLL(LIB)
lib.rg_actor_forward.argtypes = [ctypes.POINTER(_lib.RgActorWeights), ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.rg_actor_pack_gru.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
lib.rg_actor_last_error.restype = ctypes.c_char_p
_lib._lib = lib     # the actor-only diagnostic build stands in for the library
print(LIB)
N, D = 4, 16
for E, H in ((4096, 128), (4096, 64), (1024, 128)):
    actor = BatchedActor(_random_actor(1, D + N, H, 5, True, 3), N, device="cuda:0")
    obs = torch.rand(E, N, D, device="cuda:0")
    hidden = torch.zeros(E, N, H, device="cuda:0")
    waves = (E * N // 32) * (H // 32)
    q = torch.zeros(E * N * 5 + waves * 8, device="cuda:0")   # the stamps build writes its stamps behind the q block
    for _ in range(5):
        actor.forward_fused(obs, hidden, q_out=q)
    torch.cuda.synchronize()
    st = q.view(torch.int32).flatten()[E * N * 5:E * N * 5 + waves * 8].cpu().numpy().reshape(waves, 8)[:, :7].astype(np.float64)
    start = q.view(torch.int32).flatten()[E * N * 5:E * N * 5 + waves * 8].cpu().numpy().reshape(waves, 8)[:, 7].astype(np.int64)
    start = (start - start.min()) & 0x7FFFFFFF
    print(f"hipOccupancyMaxActiveBlocksPerMultiprocessor: {lib.rg_actor_occupancy(H)} workgroups per CU; wave start times (ticks after the first): "
          f"median {np.median(start):.0f}, 75 % {np.percentile(start, 75):.0f}, max {start.max()}; started within 5 k ticks: {(start < 5000).mean():.2f}")
    raw = q.view(torch.int32).flatten()[E * N * 5:E * N * 5 + waves * 8].cpu().numpy().reshape(waves, 8).astype(np.int64)
    wpt = H // 32
    xcd = (np.arange(waves) // wpt) % 8                     # workgroup b runs on XCD b % 8; every XCD has its own counter
    spans = []
    for x in range(8):
        s0 = raw[xcd == x, 7]
        s0 = (s0 - s0.min()) & 0x7FFFFFFF
        spans.append(int((s0 + raw[xcd == x, 6]).max()))
        late = float((s0 > 5000).mean())
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(50):
        actor.forward_fused(obs, hidden, q_out=q)
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / 50
    print(f"per-XCD span first wave start -> last wave end: {min(spans)} .. {max(spans)} ticks; kernel {us:.1f} us per launch -> {max(spans) / us / 1e3:.2f} ticks per ns; "
          f"waves of XCD 7 starting > 5 k ticks after its first: {late:.2f}")
    names = ["staged", "fc1", "gru mfma", "gates", "hidden stored", "fc2", "argmax/q"]
    d = np.diff(np.concatenate([np.zeros((waves, 1)), st], axis=1), axis=1)
    print(f"E {E} H {H}: {waves} waves; s_memtime ticks (100 MHz) per phase, mean over waves [mean of wave 0 of each tile]:")
    for i, n in enumerate(names):
        print(f"   {n:14s} {d[:, i].mean():8.1f}   [{d[::H // 32, i].mean():8.1f}]   cumulative {st[:, i].mean():8.1f}")
