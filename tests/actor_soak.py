#!/usr/bin/env python3
"""The actor kernel's shape fuzz (tests/test_gpu_actor.py::test_fused_actor_random_shapes) over many more draws than the test
tier runs: every input width 1 ... 64 (staged and direct fc1), 1 ... 32 actions, ragged and tiny batches, shared / per-agent
weights, the three GRU weight forms and the MLP layer, against the torch evaluation (1e-5 on q and hidden, greedy actions
equal where the top two values are 1e-4 apart).
    python tests/actor_soak.py [draws=2000] [first_seed=1000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_actor as T  # noqa: E402

draws = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t0 = time.time()
fn = getattr(T.test_fused_actor_random_shapes, "__wrapped__", T.test_fused_actor_random_shapes)
for i in range(draws):
    fn(first + i)
    if (i + 1) % 250 == 0:
        print(f"{i + 1} draws ok ({time.time() - t0:.0f} s)", flush=True)
print(f"actor soak: {draws} draws from seed {first}: all within 1e-5, greedy actions equal")
