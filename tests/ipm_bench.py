#!/usr/bin/env python3
"""csrc/ipm_qp.h alone on the GPU: wave cycles per QP and per interior-point iteration for N = 2 .. 8, one instance per lane group
(the lane-group kernel's usage) and one per lane (thread-per-env), checked bit for bit against the CPU twin
(oracle_core.h barrier_qp_ipm_spec through orc_ipm_spec_f32io).  Build: see tools/ubench/ipm_bench.hip.

    python tests/ipm_bench.py [--lib tools/ubench/libipm_bench.so] [--inst 1024]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from marbler_amd import load_config, make_params  # noqa: E402
from oracle import c_oracle  # noqa: E402  (lives under tests/: the oracle is its checker)


def draw_instances(N, count, rng, close):
    """Records (xi_x, xi_y, uhat_x, uhat_y): robots spread over the arena or clustered, position-controller inputs towards random goals."""
    io = np.zeros((count, 8, 4), np.float32)
    for t in range(count):
        while True:
            P = np.stack([rng.uniform(-1.4, 1.4, N), rng.uniform(-0.9, 0.9, N)])
            if close and t % 2:
                c = rng.uniform(-1, 1, 2) * [1.0, 0.6]
                P = c[:, None] + rng.uniform(-0.35, 0.35, (2, N))
            d = np.linalg.norm(P[:, :, None] - P[:, None, :], axis=0) + np.eye(N) * 9
            if d.min() > 0.12:
                break
        G = P + rng.choice([-1, 0, 1], (2, N)) * 0.2
        u = G - P
        nrm = np.linalg.norm(u, axis=0)
        u = np.where(nrm > 0.15, u * 0.15 / np.maximum(nrm, 1e-9), u)
        io[t, :N, 0], io[t, :N, 1], io[t, :N, 2], io[t, :N, 3] = P[0], P[1], u[0], u[1]
    return io


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "tools", "ubench", "libipm_bench.so"))
    ap.add_argument("--inst", type=int, default=1024)
    ap.add_argument("--ns", default="2,3,4,5,6,7,8")
    args = ap.parse_args()
    lib = C.CDLL(args.lib)
    olib = c_oracle.lib()
    rng = np.random.RandomState(3)
    for N in [int(v) for v in args.ns.split(",")]:
        scen, ov = ("Warehouse", {"n_agents": N, "barrier_solver": "cvxopt"})
        cfg = load_config(scen, overrides=ov)
        p = make_params(scen, cfg)
        op = c_oracle.params_from_config(scen, cfg, dtype=np.float32)
        io = draw_instances(N, args.inst, rng, True)
        want = io.copy()
        want_it = np.zeros(args.inst, np.int32)
        for t in range(args.inst):
            rec = np.ascontiguousarray(want[t, :N].reshape(-1))
            want_it[t] = olib.orc_ipm_spec_f32io(C.byref(op), N, rec.ctypes.data_as(C.POINTER(C.c_float)))
            want[t, :N] = rec.reshape(N, 4)
        for per_lane in ((0, 1) if N <= 5 else (0,)):
            d_io = torch.as_tensor(io).cuda()
            d_it = torch.zeros(args.inst, dtype=torch.int32, device="cuda")
            per_wave = 64 if per_lane else 8
            waves = (args.inst + per_wave - 1) // per_wave
            d_t = torch.zeros(waves, dtype=torch.int64, device="cuda")
            for rep in range(2):
                d_io.copy_(torch.as_tensor(io))
                rc = lib.ipm_run(C.byref(p), C.c_void_p(d_io.data_ptr()), C.c_void_p(d_it.data_ptr()), C.c_void_p(d_t.data_ptr()), N, args.inst, per_lane)
                assert rc == 0, rc
            got, got_it, ticks = d_io.cpu().numpy(), d_it.cpu().numpy(), d_t.cpu().numpy()
            same = bool(np.array_equal(got[:, :N].view(np.uint32), want[:, :N].view(np.uint32)) and np.array_equal(got_it, want_it))
            wave_max_it = np.array([want_it[w * per_wave:(w + 1) * per_wave].max() for w in range(waves)])
            if hasattr(lib, "ipm_read_ticks"):     # -DRG_IPM_STAMPS build: wave 0's ticks per phase, summed over its QP
                tk = (C.c_ulonglong * 8)()
                lib.ipm_read_ticks(tk, 1)
                its = int(want_it[:per_wave].max())
                print("  phases (ticks per iteration of wave 0: rows, residuals, assembly, factorisation, rhs+solve x2, ds/dz x2, step x2, update):",
                      [int(v / max(its, 1) / 2) for v in tk], "iterations", its)
            print(json.dumps({"N": N, "per_lane": per_lane, "instances": args.inst, "bit_exact_vs_cpu_twin": same,
                              "iterations_mean": round(float(want_it.mean()), 2), "iterations_max": int(want_it.max()),
                              "ticks_per_qp_wave_mean": int(ticks.mean()), "ticks_per_wave_iteration": int((ticks / (wave_max_it + 1.0)).mean())}), flush=True)
