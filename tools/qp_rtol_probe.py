#!/usr/bin/env python3
"""What the barrier QP's stopping tolerance costs: the headline launch (PredatorCapturePrey 4096 x 5) at several `qp_rtol`
(sim_spec_v0 ships 1.25e-6, float32-exact; the reference's cvxopt runs at reltol 1e-2).
    python tools/qp_rtol_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marbler_amd import VecRobotariumEnv
for scn, ov, nact, E in (("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 4096),
                         ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20, 2048)):
    for rtol in (1.25e-6, 2.5e-6, 5e-6, 1e-5, 1e-4, 1e-3, 1e-2):
        env = VecRobotariumEnv(scn, E, overrides=dict(ov, qp_rtol=rtol), seed=0, collect_qp_stats=True)
        acts = torch.randint(0, nact, (64, E, env.N), device=env.device, dtype=torch.int32)
        ptrs = [acts[i].data_ptr() for i in range(64)]
        env.reset()
        for i in range(300):
            env.step_raw(ptrs[i % 64])
        torch.cuda.synchronize()
        best = 1e9
        mx = 0
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(400):
                env.step_raw(ptrs[i % 64])
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 400 * 1e3)
            mx = max(mx, int(env.qp_sweeps.max()))
        print(f"{scn} {E} x {env.N} qp_rtol {rtol:8.2e}: {best:6.2f} us per step; max sweeps seen in a QP {mx}; mean {float(env.qp_sweeps.float().mean()):.3f}", flush=True)
        env.close()
