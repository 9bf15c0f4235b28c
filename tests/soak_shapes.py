#!/usr/bin/env python3
"""Long bit-exactness soak at BASELINE.json's own batch shapes (the lane -> env maps the bench lines time): thousands of
free-running steps with auto-reset against the float32 oracle on the host (16 threads), every output and state word of every
step.    python tests/soak_shapes.py [steps] [--ipm]      (--ipm: the same shapes with barrier_solver: cvxopt; the 65 536-env shape is
left out -- its CPU twin alone would take minutes per step)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # (this file lives in tests/: only tests may use the oracle)
import test_gpu_baseline_shapes as T
from oracle import c_oracle
c_oracle.build_library()
ipm = "--ipm" in sys.argv
argv = [a for a in sys.argv if a != "--ipm"]
steps = int(argv[1]) if len(argv) > 1 else 2000
total = 0
for name, scenario, ov, n_act, E, _, kernel, slots in T.CASES:
    if name not in ("pcp-4096x5-headline", "pcp-2048x5", "pcp-32768x5", "pcp-4095x5-ragged", "warehouse-4096x8", "mt-2048x6", "mt-4096x6",
                    "pcp-65536x5-auto"):
        continue
    if ipm:
        if E > 32768:
            continue
        ov = dict(ov, barrier_solver="cvxopt")
        slots = None
    if kernel is None:
        os.environ.pop("RG_STEP_KERNEL", None)
    else:
        os.environ["RG_STEP_KERNEL"] = kernel
    n = steps if scenario != "MaterialTransport" else steps // 3
    n = n if E <= 8192 else max(200, n * 8192 // E)
    t0 = time.time()
    r = T.shape_rollout_vs_oracle(name, scenario, ov, n_act, E, n, slots, c_oracle, check_rollout=False, seed=4321, action_seed=99)
    total += r["env_steps"]
    print(f"{name:24s}: {n} steps x {E} envs = {r['env_steps']} env steps, {r['episodes']} episodes, {r['violations']} violations: bit-exact "
          f"({time.time() - t0:.1f} s)", flush=True)
print("total", total, "env steps")
