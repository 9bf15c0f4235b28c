#!/usr/bin/env python3
"""Warehouse-v0 with 8 agents at a chip-filling batch (524288 envs) on both step kernels: launch time by HIP events; run it
under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE` for the instruction
counts behind DESIGN.md's costing of a two-lanes-per-env mapping for N = 7, 8 (the thread-per-env kernel's N = 8
instantiation is the one-lane program such a mapping would split in two)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from marbler_amd import VecRobotariumEnv  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
for scn, ov in (("Warehouse", {"n_agents": 8}), ("Warehouse", {"n_agents": 7}), ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5})):
    for kern in ("group", "tpe"):
        os.environ["RG_STEP_KERNEL"] = kern
        env = VecRobotariumEnv(scn, E, overrides=ov, seed=3)
        acts = torch.randint(0, 5, (8, E, env.N), device=env.device, dtype=torch.int32)
        env.reset()
        for i in range(30):
            env.step_raw(acts[i % 8].data_ptr())
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(30):
            env.step_raw(acts[i % 8].data_ptr())
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 30
        print(json.dumps({"scenario": scn, "N": env.N, "E": E, "kernel": kern, "us_per_step": us,
                          "agent_steps_per_s": E * env.N / (us * 1e-6)}), flush=True)
        env.close()
