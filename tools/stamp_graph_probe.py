#!/usr/bin/env python3
"""Phase stamps of the headline launch when it is launched by the host, step by step, and when it is the last node of a
replayed hipGraph of 20 launches (-DRG_STAMPS build): which phase of a wave is shorter in a replayed launch?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ROBOGYM_LIB"] = os.environ.get("RG_STAMPS_LIB") or os.path.join(ROOT, "marbler_amd", "librobogym_stamps.so")
sys.path.insert(0, ROOT)
import torch
from marbler_amd import VecRobotariumEnv
E, K = 4096, 20
env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, collect_qp_stats=True)
dev = env.device
acts = torch.randint(0, 5, (K, E, env.N), device=dev, dtype=torch.int32)
ptrs = [acts[i].data_ptr() for i in range(K)]
env.reset()
names = ["loaded", "ctrl1", "period1", "periods", "epilogue", "stored", "reset"]

def collect(run, reps=40):
    acc = torch.zeros(8, dtype=torch.float64); end = []
    for r in range(reps):
        run()
        torch.cuda.synchronize(dev)
        s = env.qp_sweeps.view(-1, 8).double().cpu()
        acc += s.mean(0); end.append(float(s[:, 6].max()))
    acc /= reps
    return acc, sum(end) / len(end)

def steps():
    for i in range(K):
        env.step_raw(ptrs[i])
for _ in range(5): steps()
a_s, e_s = collect(steps)
side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream(dev))
prev = env._stream; env.set_stream(side)
with torch.cuda.stream(side):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        steps()
    a_g, e_g = collect(g.replay)
env.set_stream(prev)
p0 = p1 = 0.0
for k in range(7):
    print(f"{names[k]:10s} step by step: cum {a_s[k]:8.0f} delta {a_s[k]-p0:7.0f}    graph replay: cum {a_g[k]:8.0f} delta {a_g[k]-p1:7.0f}")
    p0, p1 = a_s[k], a_g[k]
print(f"slowest wave's end (mean over launches): step by step {e_s:.0f}, graph replay {e_g:.0f} ticks")
