#!/usr/bin/env python3
"""A 20-step timed region (the driver's bench flags) launched step by step versus replayed as one hipGraph of 20 rg_step
nodes: distribution of the region's wall time over many repetitions (is the occasional slow region host-side?).
    python tools/graph_region_probe.py [regions]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marbler_amd import VecRobotariumEnv
R = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = 20
env = VecRobotariumEnv("PredatorCapturePrey", 4096, overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=0)
dev = env.device
acts = torch.randint(0, 5, (K, 4096, env.N), device=dev, dtype=torch.int32)
ptrs = [acts[i].data_ptr() for i in range(K)]
env.reset()
for i in range(50):
    env.step_raw(ptrs[i % K])
torch.cuda.synchronize()

def stats(v):
    v = sorted(v); n = len(v)
    return {"min": v[0], "median": v[n // 2], "p90": v[int(0.9 * n)], "p99": v[int(0.99 * n)], "max": v[-1]}

plain = []
for r in range(R):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(K):
        env.step_raw(ptrs[i])
    torch.cuda.synchronize(dev)
    plain.append((time.perf_counter() - t0) / K * 1e6)
    time.sleep(0.002)
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
prev = env._stream
env.set_stream(side)
with torch.cuda.stream(side):
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for i in range(K):
            env.step_raw(ptrs[i])
    graph.replay()
    torch.cuda.synchronize(dev)
    g = []
    for r in range(R):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        graph.replay()
        torch.cuda.synchronize(dev)
        g.append((time.perf_counter() - t0) / K * 1e6)
        time.sleep(0.002)
env.set_stream(prev)
print(json.dumps({"us_per_step_of_a_20_step_region": {"step_by_step": stats(plain), "one_graph_of_20_nodes": stats(g)}, "regions": R}))
