#!/usr/bin/env python3
"""Long bit-exactness soak: free-running rollouts on the GPU against the float32 oracle on the CPU
(the machinery of tests/test_gpu_rollout.py with many more steps), every scenario, both step kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # (this file lives in tests/: only tests may use the oracle)
import test_gpu_rollout as T
from oracle import c_oracle
c_oracle.build_library()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 256
for kern in ("group", "tpe"):
    os.environ["RG_STEP_KERNEL"] = kern
    for scenario, ov, n_act, _ in T.CASES:
        t0 = time.time()
        n = steps if scenario != "MaterialTransport" else steps // 3   # 74 sub-steps per step: a third as many steps
        T._rollout_bit_exact(scenario, ov, n_act, n, c_oracle, E)
        print(f"{kern:5s} {scenario:20s} {ov} : {n} steps x {E} envs bit-exact ({time.time() - t0:.1f} s)", flush=True)
