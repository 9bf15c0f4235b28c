import os, sys
sys.path.insert(0, os.getcwd())
import torch
from marbler_amd import VecRobotariumEnv
os.environ["RG_STEP_KERNEL"] = "group"
for E in (1, 8, 64, 256, 1024, 4096):
    for ar in (True,):
        env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=1, auto_reset=ar, collect_qp_stats=True)
        acts = torch.randint(0, 5, (64, E, 5), device=env.device, dtype=torch.int32)
        env.reset()
        for i in range(200): env.step_raw(acts[i % 64].data_ptr())
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(2000): env.step_raw(acts[i % 64].data_ptr())
        b.record(); torch.cuda.synchronize()
        print(E, "envs: %.2f us per step" % (a.elapsed_time(b) * 1e3 / 2000), flush=True)
# an empty-ish kernel for reference: rg_get_obs of one env (loads, epilogue, stores; no sub-step loop, no QP)
env = VecRobotariumEnv("PredatorCapturePrey", 1, overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=1)
env.reset()
for i in range(100): env.get_obs()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(2000): env.get_obs()
b.record(); torch.cuda.synchronize()
print("rg_get_obs, 1 env: %.2f us per launch" % (a.elapsed_time(b) * 1e3 / 2000))
x = torch.zeros(64, device="cuda")
for i in range(100): x.add_(1)
torch.cuda.synchronize()
a.record()
for i in range(2000): x.add_(1)
b.record(); torch.cuda.synchronize()
print("torch add_ on 64 floats: %.2f us per launch" % (a.elapsed_time(b) * 1e3 / 2000))
