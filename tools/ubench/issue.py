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

# ---- packed f32 against scalar f32, independent streams, 1 / 2 / 4 waves per SIMD (round 5; `issue.py` prints both tables)
lib.run_pk.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out_pk = torch.zeros(1024, device="cuda")
cyc_pk = torch.zeros(16, dtype=torch.int64, device="cuda")
pk_names = ["16 x v_fma_f32 (8 x/y pairs)", "8 x v_pk_fma_f32 (same arithmetic)", "8 x v_mul_f32 + 8 x v_add_f32 (4 pairs)",
            "4 x v_pk_mul_f32 + 4 x v_pk_add_f32 (same arithmetic)"]
pk_insts = [16, 8, 16, 8]           # wave instructions per trip
pk_flop_lane = [32, 32, 16, 16]     # flop per lane per trip
rows = {}
print("\npacked vs scalar f32, independent streams; flop / clk / SIMD (gfx950 peak as specified: 64 = 157.3 TF / 1024 SIMDs / 2.4 GHz)")
for wps in (1, 2, 4):
    for mode in range(4):
        best = None
        for _ in range(5):
            cyc_pk.zero_()
            assert lib.run_pk(mode, wps, out_pk.data_ptr(), cyc_pk.data_ptr()) == 0
            torch.cuda.synchronize()
            t = int(cyc_pk[: 4 * wps].max())
            best = t if best is None else min(best, t)
        trips = 256
        flop_clk_simd = wps * trips * pk_flop_lane[mode] * 64 / best
        cyc_per_inst = best / (trips * pk_insts[mode] * wps)
        rows[(wps, mode)] = flop_clk_simd
        print(f"waves/SIMD {wps}  {pk_names[mode]:52s} {best:8d} ticks  {cyc_per_inst:5.2f} cyc per wave-instruction per wave  "
              f"{flop_clk_simd:6.1f} flop/clk/SIMD", flush=True)
for wps in (1, 2, 4):
    print(f"waves/SIMD {wps}: packed / scalar  fma {rows[(wps, 1)] / rows[(wps, 0)]:.3f}   mul+add {rows[(wps, 3)] / rows[(wps, 2)]:.3f}")
