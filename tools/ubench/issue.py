#!/usr/bin/env python3
"""Driver of tools/ubench/issue.hip.  Build first:
   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/ubench/libissue.so tools/ubench/issue.hip"""
import ctypes, os
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libissue.so"))
lib.run_mode.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(64, device="cuda")
cyc = torch.zeros(8, dtype=torch.int64, device="cuda")
names = ["dependent v_fma chain", "2 independent v_fma chains", "4 independent v_fma chains",
         "s_nop 1 + v_mov_dpp -> v_fma chain (per pair of instr)", "dependent v_rcp chain", "v_cmp -> v_cndmask chain (per pair)",
         "dependent v_fma + 1 independent v_mul (per pair)", "dependent v_fma + 3 independent v_mul (per 4)"]
per = [1, 1, 1, 2, 1, 2, 2, 4]
for mode in range(8):
    for act in (64, 16):
        for _ in range(3):
            lib.run_mode(mode, act, out.data_ptr(), cyc.data_ptr())
            torch.cuda.synchronize()
        n = 128 * 16 / per[mode]
        print(f"{names[mode]:56s} active {act:2d}: {int(cyc[0]):7d} ticks, {int(cyc[0]) / n:.2f} per unit", flush=True)
