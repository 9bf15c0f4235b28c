"""The oracle against the reference's own vectors (CPU, no GPU).

tests/golden/*.npz were captured by running the reference's Python (Wrapper -> scenario -> roboEnv
-> Controller) over the restated rps; they pin rows a1, a2, a11-a16 of SURVEY.md section 8.
 - tier 1, oracle/np_port.py (NumPy float64, the reference's shape): float64-EQUAL, free-running
   whole episodes including resets (same NumPy RNG stream).
 - tier 2, oracle/oracle.c float64: teacher-forced per step, masks exact, values to 1e-12.
"""
import os

import numpy as np
import pytest

from helpers import angle_diff, golden_files, load_golden, oracle_from_state, pre_state

MSG = {"": 0, "collision": 1, "boundary": 2, "collision_boundary": 3}


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_c_oracle_f64_matches_reference_vectors(path, oracle_lib):
    g, scenario, cfg = load_golden(path)
    env = oracle_from_state(oracle_lib, scenario, cfg, pre_state(g), np.float64)
    obs, rew, done, info = env.step(g["actions"])
    # exact-projection fixtures: the C tier repeats the Python operation for operation (1e-12).  ipm_* fixtures: two restatements of
    # cvxopt's iteration that order the linear algebra differently (numpy / LAPACK Cholesky in oracle/rps_restated/cvxopt_restated.py,
    # plain loops in oracle_core.h barrier_qp_ipm) -- rounding-level differences, amplified by the controller (measured max 4e-12)
    tol = 1e-9 if os.path.basename(path).startswith("ipm_") else 1e-12
    assert np.array_equal(done, g["done"])
    assert np.array_equal(info["violation"], g["viol"])
    assert np.array_equal(info["remaining"], g["remaining"])
    assert np.abs(obs - g["obs"]).max() < tol
    assert np.abs(rew - g["reward"]).max() < tol
    assert np.abs(info["dist_travelled"] - g["dist"]).max() < tol
    assert np.abs(env.poses[:, :2] - g["post_poses"][:, :2]).max() < tol
    assert angle_diff(env.poses[:, 2], g["post_poses"][:, 2]).max() < tol
    assert np.abs(env.carry - g["post_carry"]).max() < tol
    assert np.array_equal(env.steps, g["post_steps"])
    for k in ("prey_sensed", "prey_captured", "loaded", "load", "zone_load", "messages", "pixel_type", "reached_goal"):
        if "post_" + k in g.files:
            assert np.array_equal(getattr(env, k).reshape(g["post_" + k].shape), g["post_" + k]), k


@pytest.mark.parametrize("path", [p for p in golden_files() if "viol_" not in p], ids=lambda p: p.split("/")[-1][:-4])
def test_numpy_port_equals_reference_vectors(path):
    from oracle import np_port
    import rps.robotarium as rr
    g, scenario, cfg = load_golden(path)
    per = int(g["steps_per_seed"])
    import random
    for si, seed in enumerate(g["seeds"]):
        rr._ERRORS.clear()
        random.seed(int(seed) + 12345)     # ArcticTransport draws its goal column from Python's `random`
        port = np_port.make_port(scenario, dict(cfg, seed=int(seed)))
        port.reset()
        for t in range(si * per, (si + 1) * per):
            assert np.array_equal(port.agent_poses, g["pre_poses"][t]), t
            obs, rew, done, info = port.step([int(a) for a in g["actions"][t]])
            assert np.array_equal(np.array(obs), g["obs"][t]), t
            assert np.array_equal(np.array(rew, dtype=np.float64), g["reward"][t]), t
            assert bool(done[0]) == bool(g["done"][t]), t
            assert np.array_equal(info["dist_travelled"], g["dist"][t]), t
            assert MSG[info.get("message", "")] == int(g["viol"][t]), t
            assert info.get("remaining", -1) == int(g["remaining"][t]), t
            if done[0]:
                port.reset()


def test_fixtures_cover_the_edge_cases():
    """The vectors exercise what the reference's semantics hinge on."""
    seen = {"capture": False, "sense": False, "timeout": False, "viol": set(), "load": False, "unload": False,
            "mt_done_empty": False, "knn": False, "all_others": False}
    for path in golden_files():
        g, scenario, cfg = load_golden(path)
        seen["viol"] |= set(int(v) for v in g["viol"])
        if scenario == "PredatorCapturePrey":
            seen["capture"] |= bool((g["post_prey_captured"].sum(1) > g["pre_prey_captured"].sum(1)).any())
            seen["sense"] |= bool((g["post_prey_sensed"].sum(1) > g["pre_prey_sensed"].sum(1)).any())
            n = cfg["predator"] + cfg["capture"]
            seen["knn" if cfg["num_neighbors"] < n - 1 else "all_others"] = True
            seen["timeout"] |= bool(((g["done"] == 1) & (g["viol"] == 0) & (g["remaining"] > 0)).any())
        if scenario == "Warehouse":
            seen["load"] |= bool((g["post_loaded"] > g["pre_loaded"]).any())
            seen["unload"] |= bool((g["post_loaded"] < g["pre_loaded"]).any())
        if scenario == "MaterialTransport":
            seen["mt_done_empty"] |= bool(((g["done"] == 1) & (g["remaining"] == 0)).any())
    assert seen["viol"] >= {0, 1, 2, 3}
    for k in ("capture", "sense", "timeout", "load", "unload", "knn", "all_others"):
        assert seen[k], k
