import os, sys
sys.path.insert(0, os.getcwd())
os.environ["RG_STEP_KERNEL"] = "group"
import torch
from marbler_amd import VecRobotariumEnv
CASES = [("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, (4096, 16384, 32768, 65536)),
         ("Warehouse", {"n_agents": 8}, 5, (4096, 16384, 32768, 65536, 131072)),
         ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20, (2048, 4096, 16384, 32768))]
for scn, ov, nact, Es in CASES:
    for E in Es:
        env = VecRobotariumEnv(scn, E, overrides=ov, seed=0)
        acts = torch.randint(0, nact, (16, E, env.N), device=env.device, dtype=torch.int32)
        ptrs = [acts[i].data_ptr() for i in range(16)]
        env.reset()
        for i in range(100):
            env.step_raw(ptrs[i % 16])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(300):
            env.step_raw(ptrs[i % 16])
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 300 * 1e3
        print(f"group {scn:20s} N={env.N} E={E:7d} {us:8.2f} us/step", flush=True)
        env.close()
