"""Shared test helpers: golden fixtures, oracle envs, GPU envs loaded from the same state."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STATE_KEYS = ("poses", "carry", "steps", "prey_loc", "prey_sensed", "prey_captured", "loaded", "load",
              "zone_load", "messages", "grid", "goal_col", "pixel_type", "reached_goal")
GPU_NAME = {"carry": "carry_dist", "steps": "episode_steps"}
FLAT = ("goal_col",)


def golden_files():
    """Step fixtures (the actor_*.npz files are the policy-network vectors of tests/test_evaluate.py, zoo_eval_replay.npz the
    recorded policy rollouts of tests/test_zoo_eval.py)."""
    return sorted(f for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")) if not os.path.basename(f).startswith(("actor_", "zoo_")))


def load_golden(path):
    g = np.load(path)
    return g, str(g["scenario"]), json.loads(str(g["config_json"]))


def pre_state(g):
    return {k: g["pre_" + k] for k in STATE_KEYS if "pre_" + k in g.files}


def oracle_from_state(c_oracle, scenario, cfg, state, dtype):
    """One oracle env per fixture row, loaded with the row's pre-step state."""
    T = len(state["poses"])
    env = c_oracle.OracleVecEnv(scenario, cfg, T, dtype=dtype)
    for k, v in state.items():
        arr = getattr(env, k)
        arr[...] = np.asarray(v).astype(arr.dtype).reshape(arr.shape)
    return env


def gpu_from_state(scenario, cfg, state, **kw):
    import torch
    from marbler_amd import VecRobotariumEnv
    T = len(state["poses"])
    env = VecRobotariumEnv(scenario, T, overrides=cfg, auto_reset=False, collect_qp_stats=True, **kw)
    sd = {GPU_NAME.get(k, k): torch.as_tensor(np.asarray(v)) for k, v in state.items()}
    env.load_state_dict(sd)
    return env


def angle_diff(a, b):
    return np.abs(np.angle(np.exp(1j * (np.asarray(a, np.float64) - np.asarray(b, np.float64)))))


def oracle_reset_params(c_oracle, rg_params):
    """orc_reset_params from the product's rg_scenario_params (same float values)."""
    rp = c_oracle.OrcResetParams()
    rp.scenario, rp.n_agents, rp.num_prey, rp.keep_theta = \
        rg_params.scenario, rg_params.n_agents, rg_params.num_prey, rg_params.keep_theta
    for name in ("agent_grid", "prey_grid"):
        src, dst = getattr(rg_params, name), getattr(rp, name)
        for f, _ in c_oracle.OrcGrid._fields_:
            setattr(dst, f, getattr(src, f))
    rp.zone1_mean, rp.zone1_std = rg_params.zone1_mean, rg_params.zone1_std
    rp.zone2_mean, rp.zone2_std = rg_params.zone2_mean, rg_params.zone2_std
    return rp


# the oracle's twin of the device reset sampler, written into row e of an OracleVecEnv (auto-reset of a finished env)
def oracle_reset(oracle_lib, orc, rp, seed, e, episode):
    if orc.scenario == "ArcticTransport":
        p, grid, gc = oracle_lib.reset_arctic_f32(seed, e, episode)
        orc.poses[e] = p
        orc.carry[e] = 0
        orc.steps[e] = 0
        orc.grid[e] = grid
        orc.goal_col[e] = gc
        orc.pixel_type[e] = 0
        orc.reached_goal[e] = 0
        return
    p, q, z = oracle_lib.reset_env_f32(rp, seed, e, episode)
    orc.poses[e] = p
    orc.carry[e] = 0
    orc.steps[e] = 0
    orc.prey_loc[e] = q[:orc.prey_loc.shape[1]]
    orc.prey_sensed[e] = 0
    orc.prey_captured[e] = 0
    orc.loaded[e] = 0
    orc.load[e] = 0
    if orc.scenario == "MaterialTransport":
        orc.zone_load[e] = z
        orc.messages[e] = 0
