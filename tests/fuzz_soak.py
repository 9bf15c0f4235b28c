#!/usr/bin/env python3
"""One-off fuzz campaign on a GPU box: draws [first, first + count) of tests/test_gpu_config_fuzz.py's EXTENDED generator,
each a free-running rollout of the HIP kernels against the float32 oracle, bit for bit.
    python tests/fuzz_soak.py [first] [count] [seconds] [--ipm]   (stops at the time budget; prints one summary line)
--ipm: the draws of test_gpu_config_fuzz.draw_interior_point instead (barrier_solver: cvxopt, n_agents <= 8, certificate family and
cvxopt's own options drawn too)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    ipm = "--ipm" in sys.argv
    argv = [a for a in sys.argv if a != "--ipm"]
    first = int(argv[1]) if len(argv) > 1 else 100000
    count = int(argv[2]) if len(argv) > 2 else 2000
    budget = float(argv[3]) if len(argv) > 3 else 600.0
    from oracle import c_oracle
    c_oracle.build_library()
    from test_gpu_config_fuzz import draw_config_extended, draw_interior_point
    draw = draw_interior_point if ipm else draw_config_extended
    from test_gpu_rollout import _rollout_bit_exact
    from marbler_amd.params import load_config, make_params
    t0, done, rejected, env_steps, failures = time.time(), 0, 0, 0, []
    per_scn = {}
    for i in range(first, first + count):
        if time.time() - t0 > budget:
            break
        scenario, ov, n_act, E, kernel = draw(np.random.RandomState(i))
        try:
            make_params(scenario, load_config(scenario, None, ov))
        except Exception:                       # a draw outside what the parameter block admits (e.g. a grid with too few cells)
            rejected += 1
            continue
        os.environ["RG_STEP_KERNEL"] = kernel
        steps = 30 if scenario != "MaterialTransport" else 20
        try:
            _rollout_bit_exact(scenario, ov, n_act, steps, c_oracle, E, require_done=False)
        except AssertionError as exc:
            failures.append((i, scenario, ov, E, kernel, str(exc)[:300]))
            print("FAIL", failures[-1], flush=True)
        done += 1
        env_steps += E * steps
        per_scn[scenario] = per_scn.get(scenario, 0) + 1
        if done % 200 == 0:
            print(f"{done} draws, {env_steps} env steps, {len(failures)} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_soak{' (interior-point mode)' if ipm else ''}: draws {first}..{first + count}: {done} run ({per_scn}), {rejected} rejected by make_params, "
          f"{env_steps} env steps, {len(failures)} failures, {time.time() - t0:.0f} s")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
