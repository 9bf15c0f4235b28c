import os, sys
ROOT = "/root/repo"
os.environ["ROBOGYM_LIB"] = os.path.join(os.getcwd(), "marbler_amd", "librobogym_stamps.so")
os.environ["RG_STEP_KERNEL"] = "tpe"
sys.path.insert(0, os.getcwd())
import torch
from marbler_amd import VecRobotariumEnv
E = int(sys.argv[1])
env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, collect_qp_stats=True)
acts = torch.randint(0, 5, (64, E, 5), device=env.device, dtype=torch.int32)
env.reset()
acc = torch.zeros(8, dtype=torch.float64); n = 0
for i in range(200):
    env.step(acts[i % 64])
    if i >= 100:
        s = env.qp_sweeps.view(-1, 64)[:, :8].double().cpu(); acc += s.mean(0); n += 1
acc /= n
names = ["prey done", "obs out", "last period end(2)", "periods(3)", "epilogue(4)", "stored(5)"]
order = [3, 0, 1, 4, 5]
print("E", E, {names[i]: round(float(acc[i])) for i in range(6)})
prev = acc[3]
for i in order[1:]:
    print(f"  {names[i]:14s} +{float(acc[i]-prev):8.0f}"); prev = acc[i]
