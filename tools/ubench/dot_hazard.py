#!/usr/bin/env python3
"""Driver of tools/ubench/dot_hazard.hip: how many wait states a VALU consumer of a v_dot2_f32_f16 result needs on this GPU."""
import ctypes
import os

import numpy as np
import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdot_hazard.so"))
lib.run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
rng = np.random.RandomState(0)
h = rng.uniform(-0.3, 0.3, (64, 2)).astype(np.float16)
want = (h.astype(np.float32) ** 2).sum(axis=1).astype(np.float32).view(np.int32)
inp = torch.as_tensor(h.view(np.int32).reshape(64), device="cuda")
for variant, label in ((0, "separate destination, 0 wait states"), (1, "1 wait state "), (2, "2 wait states"), (3, "3 wait states (s_nop 2)"),
                       (10, "in place (dst = src), 0 wait states"), (13, "in place, 3 wait states")):
    out = torch.full((64,), -1, dtype=torch.int32, device="cuda")
    lib.run(variant, inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ok = int((np.abs(got.view(np.float32) - want.view(np.float32)) < 1e-6).sum())
    stale = int((got == inp.cpu().numpy()).sum())
    print(f"{label:42s}: {ok:2d} of 64 lanes hold dot2(d, d); {stale:2d} hold the packed input itself; lane 0: {got[0]:#010x} (want {want[0]:#010x})")
