#!/usr/bin/env python3
"""Evaluation-loop rate (policy forward + arg-max + env step per iteration), eager vs hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from marbler_amd import VecRobotariumEnv
from marbler_amd.evaluate import BatchedActor, run_eval
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "actor_shared_gru.npz"))
sd = {k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("sd_")}
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "test_gpu_actor.py"))
t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
for E in (256, 4096):
    for H in (16, 64, 128):
        for fused in ((False,) if H == 16 else (False, True)):
            for use_graph in (False, True):
                env = VecRobotariumEnv("PredatorCapturePrey", E, seed=5)
                actor = BatchedActor(sd if H == 16 else t._random_actor(1, 20, H, 5, True, 3), env.N, device=env.device)
                run_eval(env, actor, steps=20, use_graph=use_graph, fused=fused)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = run_eval(env, actor, steps=400, use_graph=use_graph, fused=fused)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                print(f"E={E} H={H} fused={fused} graph={use_graph}: {dt / 400 * 1e6:.1f} us per iteration, "
                      f"{E * env.N * 400 / dt / 1e6:.1f} M agent-steps/s, episodes {out['episodes']}", flush=True)
