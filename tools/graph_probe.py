#!/usr/bin/env python3
"""The evaluation loop (fused actor + env step + distance sum) eager vs hipGraph replay, by GRU weight form, 4096 envs.
Round 5: eager 31 / 38 us per iteration (f16x2 / bf16x3, hidden 128), replay 38 / 58: the replay is the slower way for this loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from marbler_amd import VecRobotariumEnv
from marbler_amd.evaluate import BatchedActor, run_eval
from test_gpu_actor import _random_actor
for H in (128, 64):
    for pack in ("bf16x3", "f16x2"):
        for use_graph in (False, True):
            env = VecRobotariumEnv("PredatorCapturePrey", 4096, seed=5)
            actor = BatchedActor(_random_actor(1, 20, H, 5, True, 3), env.N, device=env.device, pack_gru=pack)
            run_eval(env, actor, steps=20, use_graph=use_graph, fused=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = run_eval(env, actor, steps=400, use_graph=use_graph, fused=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"H={H} pack={pack} graph={use_graph}: {dt / 400 * 1e6:.1f} us per iteration", flush=True)
