#!/usr/bin/env python3
"""What a step launch fetches besides its envs' data: the same kernel at 8, 64, 512 and 4096 envs (grid = 8 .. 1024
one-wave workgroups), run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`; tools/summarize_rocpd.py gives one
row per grid size.  FETCH_SIZE x 2 (profiles/r3_hbm_calibration.csv) at 8 envs -- one workgroup on each XCD, 1.5 KB of
env data -- is the launch's fixed traffic: chiefly the kernel's own instructions, which every XCD's L2 fetches again
after each kernel boundary (the L2s are written back and invalidated between launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from marbler_amd import VecRobotariumEnv  # noqa: E402

os.environ["RG_STEP_KERNEL"] = "group"
for E in (8, 64, 512, 4096):
    env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=1)
    acts = torch.randint(0, 5, (64, E, 5), device=env.device, dtype=torch.int32)
    env.reset()
    for i in range(300):
        env.step_raw(acts[i % 64].data_ptr())
    torch.cuda.synchronize()
    print(E, "envs:", (E + 0) and ((E + (1 if E <= 1024 else 2 if E <= 2048 else 4) - 1) // (1 if E <= 1024 else 2 if E <= 2048 else 4)), "workgroups")
    env.close()
