#!/usr/bin/env python3
"""A/B bench for the fused actor kernel: every tools/actor_lab/build/*.so (csrc/actor_mfma.hip alone, built with
`python tools/actor_lab/lab.py build name="-Dflag ..." ...`) is loaded on its own, checked against the torch evaluation of the
same weights (q and hidden within 1e-5) and timed at 4096 envs x 4 agents (hidden 128 and 64) and at smaller batches.
    python tools/actor_lab/lab.py build v2="" v2s="-DRG_ACTOR_STAMPS"
    python tools/actor_lab/lab.py            # on a GPU"""
import ctypes as C
import glob
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(HERE, "build")

if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for arg in sys.argv[2:]:
        name, flags = arg.split("=", 1)
        # names starting with "v1": the round-3 kernel; not kept in the tree -- git show 932219d:marbler_amd/csrc/actor_mfma.hip > tools/actor_lab/actor_v1.hip
        src = os.path.join(HERE, "actor_v1.hip") if name.startswith("v1") else os.path.join(ROOT, "marbler_amd", "csrc", "actor_mfma.hip")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "marbler_amd", "csrc"),
               "-shared", src, "-o", os.path.join(OUT, f"actor_{name}.so")] + flags.split()
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        print(name, "built" if p.wait() == 0 else "FAILED")
    sys.exit(0)

import numpy as np
import torch
from marbler_amd import _lib
from marbler_amd.evaluate import BatchedActor
from test_gpu_actor import _random_actor


def bind(path):
    lib = C.CDLL(path)
    lib.rg_actor_forward.argtypes = [C.POINTER(_lib.RgActorWeights), C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rg_actor_forward.restype = C.c_int
    lib.rg_actor_pack_gru.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rg_actor_pack_gru.restype = C.c_int
    if hasattr(lib, "rg_actor_pack_gru_bf16x3"):
        lib.rg_actor_pack_gru_bf16x3.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        lib.rg_actor_pack_gru_bf16x3.restype = C.c_int
    if hasattr(lib, "rg_actor_pack_gru_f16x2"):
        lib.rg_actor_pack_gru_f16x2.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        lib.rg_actor_pack_gru_f16x2.restype = C.c_int
    lib.rg_actor_last_error.restype = C.c_char_p
    return lib


def case(lib, E, N, D, H, A, shared=True, use_rnn=True, reps=100, check=True, pack=True):
    dev = "cuda:0"
    I = D + N
    if pack is True and not hasattr(lib, "rg_actor_pack_gru_bf16x3"):
        pack = "f32"
    actor = BatchedActor(_random_actor(1 if shared else N, I, H, A, use_rnn, 3), N, use_rnn=use_rnn, device=dev, pack_gru=pack)
    _lib._lib = lib            # BatchedActor.forward_fused / _weights_struct go through marbler_amd._lib.load()
    actor._ws = None
    g = torch.Generator(device=dev).manual_seed(1)
    obs = torch.rand(E, N, D, generator=g, device=dev) * 3 - 1.5
    hidden = torch.rand(E, N, H, generator=g, device=dev) * 2 - 1
    restart = (torch.rand(E, generator=g, device=dev) < 0.2).to(torch.uint8)
    err = None
    if check:
        eye = torch.eye(N, device=dev).unsqueeze(0).expand(E, N, N)
        fresh = restart.bool()[:, None, None]
        q_ref, h_ref = actor.forward(torch.cat([torch.where(fresh, torch.zeros_like(obs), obs), eye], dim=2), torch.where(fresh, torch.zeros_like(hidden), hidden))
        h = hidden.clone()
        q, act = actor.forward_fused(obs, h, restart=restart)
        torch.cuda.synchronize()
        top2 = q_ref.topk(2, dim=2).values
        clear = (top2[..., 0] - top2[..., 1]) > 1e-4
        err = {"q": float((q - q_ref).abs().max()), "h": float((h - h_ref).abs().max()),
               "argmax_ok": bool(torch.equal(act[clear].long(), q_ref.argmax(dim=2)[clear]))}
    q = torch.empty(E, N, A, device=dev)
    act = torch.empty(E, N, dtype=torch.int32, device=dev)
    for _ in range(20):
        actor.forward_fused(obs, hidden, q_out=q, actions_out=act)
    torch.cuda.synchronize()
    times = []
    for _ in range(7):   # the fastest of seven bursts (a burst now and then runs 5-7x slower on this pool: clock / power state)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            actor.forward_fused(obs, hidden, q_out=q, actions_out=act)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) * 1e3 / reps)
    us = min(times)
    flop = 2.0 * E * N * (I * H + (2 * 3 * H * H if use_rnn else H * H) + H * A)
    return {"pack": str(pack), "E": E, "N": N, "H": H, "A": A, "shared": shared, "use_rnn": use_rnn, "us": round(us, 2), "us_median": round(sorted(times)[len(times) // 2], 2), "tflops": round(flop / us / 1e6, 1),
            "frac_f32_mfma_peak": round(flop / us / 1e6 / 157.3, 3), "err": err}


if __name__ == "__main__":
    out = {}
    for path in sorted(glob.glob(os.path.join(OUT, "actor_*.so"))):
        name = os.path.basename(path)[6:-3]
        lib = bind(path)
        rows = []
        for (E, N, D, H, A, shared, rnn) in ((4096, 4, 16, 128, 5, True, True), (4096, 4, 16, 64, 5, True, True), (1024, 4, 16, 128, 5, True, True),
                                             (8192, 4, 16, 128, 5, True, True), (4096, 5, 16, 128, 5, False, True), (4096, 4, 9, 128, 20, True, False),
                                             (300, 5, 16, 128, 5, True, True)):
            packs = (True,)
            if rnn and hasattr(lib, "rg_actor_pack_gru_f16x2"):
                packs = ("f16x2", "bf16x3") if "--all" not in sys.argv else ("f16x2", "bf16x3", "f32")
            elif rnn and hasattr(lib, "rg_actor_pack_gru_bf16x3"):
                packs = (True, "f32")
            for pack in packs:
                rows.append(case(lib, E, N, D, H, A, shared, rnn, pack=pack))
                print(name, json.dumps(rows[-1]), flush=True)
        if hasattr(lib, "rg_actor_occupancy"):
            print(name, "workgroups per CU (runtime):", lib.rg_actor_occupancy(128), lib.rg_actor_occupancy(64), flush=True)
        out[name] = rows
    with open(os.path.join(ROOT, "gpurun_out", "actor_lab.json"), "w") as f:
        json.dump(out, f, indent=1)
