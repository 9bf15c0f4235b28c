#!/usr/bin/env python3
"""Where a wave of the fused actor kernel spends its time: a -DRG_ACTOR_STAMPS build of csrc/actor_mfma.hip alone (phase stamps
from s_memtime written behind q; python tools/actor_lab/lab.py build v2s="-DRG_ACTOR_STAMPS").
    python tools/actor_stamps.py [--lib tools/actor_lab/build/actor_v2s.so] [--pack f16x2|bf16x3|f32]"""
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
if hasattr(lib, "rg_actor_pack_gru_bf16x3"):
    lib.rg_actor_pack_gru_bf16x3.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
if hasattr(lib, "rg_actor_pack_gru_f16x2"):
    lib.rg_actor_pack_gru_f16x2.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
PACK = sys.argv[sys.argv.index("--pack") + 1] if "--pack" in sys.argv else True
_lib._lib = lib     # the actor-only diagnostic build stands in for the library
print(LIB)
N, D = 4, 16
for E, H in ((4096, 128), (4096, 64), (1024, 128)):
    actor = BatchedActor(_random_actor(1, D + N, H, 5, True, 3), N, device="cuda:0", pack_gru=PACK)
    obs = torch.rand(E, N, D, device="cuda:0")
    hidden = torch.zeros(E, N, H, device="cuda:0")
    waves = (E * N // 32) * (H // 32)
    q = torch.zeros(E * N * 5 + waves * 8, device="cuda:0")   # the stamps build writes its stamps behind the q block
    for _ in range(5):
        actor.forward_fused(obs, hidden, q_out=q)
    torch.cuda.synchronize()
    st = q.view(torch.int32).flatten()[E * N * 5:E * N * 5 + waves * 8].cpu().numpy().reshape(waves, 8)[:, :7].astype(np.float64)
    st[:, 0] = 0
    print(f"hipOccupancyMaxActiveBlocksPerMultiprocessor: {lib.rg_actor_occupancy(H)} workgroups per CU")
    raw = q.view(torch.int32).flatten()[E * N * 5:E * N * 5 + waves * 8].cpu().numpy().reshape(waves, 8).astype(np.int64)
    wpt = H // 32
    xcc, hwid = raw[:, 0] >> 16, raw[:, 0] & 0xFFFF
    cu = (hwid >> 8) & 0xFF            # CU + SH + SE bits: one id per CU of an XCC
    life = raw[:, 6]
    rt = raw[:, 7].astype(np.float64)
    print(f"XCCs seen: {sorted(set(int(v) for v in xcc))}; CUs per XCC used: {np.mean([len(np.unique(cu[xcc == x])) for x in np.unique(xcc)]):.1f}; "
          f"waves per (XCC, CU): {waves / max(1, len(set(zip(xcc.tolist(), cu.tolist())))):.2f}")
    print(f"wave life: {life.mean():.0f} shader cycles = {rt.mean() / 100:.2f} us on the 100 MHz clock -> shader clock {life.sum() / rt.sum() * 100:.0f} MHz while the waves ran")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(50):
        actor.forward_fused(obs, hidden, q_out=q)
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / 50
    print(f"kernel {us:.1f} us per launch (HIP events over 50 launches)")
    if "--fc1" in sys.argv:   # -DRG_ACTOR_STAMPS_FC1 build: the head of the wave in detail (slots 2..5, then slot 1 = behind the barrier)
        print("   cumulative ticks: hidden loads issued %.0f, fc1 addresses done %.0f, fc1 products + bias arrived %.0f, Y / Hs written %.0f, barrier passed %.0f, end %.0f"
              % tuple(raw[:, i].mean() for i in (2, 3, 4, 5, 1, 6)))
        continue
    raw[:, 0] = 0
    names = ["(where)", "fc1", "gru mfma", "gates", "hidden stored", "fc2", "argmax/q"]
    d = np.diff(np.concatenate([np.zeros((waves, 1)), st], axis=1), axis=1)
    print(f"E {E} H {H}: {waves} waves; s_memtime ticks per phase, mean over waves [mean of wave 0 of each tile]:")
    for i, n in enumerate(names):
        print(f"   {n:14s} {d[:, i].mean():8.1f}   [{d[::H // 32, i].mean():8.1f}]   cumulative {st[:, i].mean():8.1f}")
