"""tests/golden/ZOO_EVAL.json: what the reference's own trained policies score on this engine (make_zoo_eval.py), and the
recorded-action replay that ties its float32 rows to the oracle (here) and to the HIP kernels (tests/test_gpu_zoo_replay.py)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, oracle_reset, oracle_reset_params

REC = os.path.join(GOLDEN_DIR, "ZOO_EVAL.json")
REPLAY = os.path.join(GOLDEN_DIR, "zoo_eval_replay.npz")
ZOO = "/root/reference/robotarium_gym/scenarios"


def test_record_covers_the_model_zoo():
    rec = json.load(open(REC))
    assert len(rec["models"]) == 22 and rec["episodes_per_row"] >= 200
    for name, row in rec["models"].items():
        assert set(row["variants"]) == {"float32", "float64_exact", "float64_cvxopt_restated", "float32_cvxopt_restated"}, name
        f32, f64 = row["variants"]["float32"], row["variants"]["float64_exact"]
        # round 5: the interior-point mode's float32 tier (= the kernels with barrier_solver: cvxopt) against the float64 restatement
        i32, i64 = row["variants"]["float32_cvxopt_restated"], row["variants"]["float64_cvxopt_restated"]
        se_i = np.hypot(i32["return_std"], i64["return_std"]) / np.sqrt(i32["episodes"])
        assert abs(i32["return_mean"] - i64["return_mean"]) <= 5 * se_i + 1e-9, (name, i32["return_mean"], i64["return_mean"])
        assert abs(i32["steps_mean"] - i64["steps_mean"]) <= 5 * np.hypot(i32["steps_std"], i64["steps_std"]) / np.sqrt(i32["episodes"]) + 1e-9, name
        for v in row["variants"].values():
            assert v["episodes"] == rec["episodes_per_row"] and sum(v["ended_by"].values()) == v["episodes"], name
        # the float32 tier (= the kernels) against the same spec in float64, same initial states, same policy: trajectories part
        # at threshold flips, the statistics must not (5 standard errors of a difference of two correlated means, generously)
        se = np.hypot(f32["return_std"], f64["return_std"]) / np.sqrt(v["episodes"])
        assert abs(f32["return_mean"] - f64["return_mean"]) <= 5 * se + 1e-9, (name, f32["return_mean"], f64["return_mean"])
        assert abs(f32["steps_mean"] - f64["steps_mean"]) <= 5 * np.hypot(f32["steps_std"], f64["steps_std"]) / np.sqrt(v["episodes"]) + 1e-9, name
    if os.path.isdir(ZOO):   # build container: one row per checkpoint of the reference tree
        import glob
        assert len(glob.glob(os.path.join(ZOO, "*", "models", "*.th"))) == len(rec["models"])


def test_replay_through_the_float32_oracle_reproduces_the_recorded_statistics(oracle_lib):
    """The recorded actions of the three replay rows through the float32 oracle from the sampler twin's initial states: the
    per-episode statistics of the npz, exactly (what the GPU tier then asks of the kernels)."""
    from marbler_amd.params import load_config, make_params
    z = np.load(REPLAY)
    for row, prefix, ov in [(str(r), pre, ov) for r in z["rows"] for pre, ov in (("", {}), ("ipm__", {"barrier_solver": "cvxopt"}))]:
        key, scenario = prefix + row.replace("/", "__"), row.split("/")[0]
        acts = z[f"{key}__actions"]
        T, E, N = acts.shape
        cfg = load_config(scenario, overrides=ov or None)
        p = make_params(scenario, cfg)
        orc = oracle_lib.OracleVecEnv(scenario, cfg, E, dtype=np.float32)
        rp = oracle_reset_params(oracle_lib, p)
        for e in range(E):
            oracle_reset(oracle_lib, orc, rp, int(z["seed"]), e, 0)
        ret, dist = np.zeros(E, np.float64), np.zeros((E, N), np.float64)
        steps, active = np.zeros(E, np.int64), np.ones(E, bool)
        for j in range(T):
            _, r, d, info = orc.step(acts[j].astype(np.int32))
            rr = r.astype(np.float64)
            ret[active] += (rr[:, 0] if p.shared_reward else rr.sum(axis=1))[active]
            dist[active] += info["dist_travelled"].astype(np.float64)[active]
            ended = active & (d != 0)
            steps[ended] = j + 1
            active &= ~ended
        assert not active.any(), row
        assert np.array_equal(steps, z[f"{key}__steps"]) and np.array_equal(ret, z[f"{key}__return"]) and np.array_equal(dist, z[f"{key}__dist"]), row


@pytest.mark.skipif(not os.path.isdir(ZOO), reason="the reference tree (with its model zoo) is not on this machine")
def test_recorded_actions_are_what_the_reference_policy_chooses():
    """Build container only: the reference's own module + checkpoint, re-evaluated for one replay row, picks the recorded
    actions (the generator is reproducible: same machine, same torch)."""
    import sys
    sys.path.insert(0, GOLDEN_DIR)
    import make_zoo_eval as mz
    from marbler_amd.gymma import N_ACTIONS
    from marbler_amd.params import load_config, make_params
    z = np.load(REPLAY)
    row = str(z["rows"][0])
    scenario, model_name = row.split("/")
    cfg = load_config(scenario)
    p = make_params(scenario, cfg)
    mdir = os.path.join(ZOO, scenario, "models")
    model, mcfg, ns, _ = mz.load_reference_actor(os.path.join(mdir, model_name + ".th"), os.path.join(mdir, model_name + ".json"), p.n_agents, N_ACTIONS[scenario])
    _, replay = mz.evaluate(scenario, cfg, p, model, mcfg, ns, mz.REPLAY_ENVS, np.float32, record=mz.REPLAY_ENVS)
    key = row.replace("/", "__")
    assert np.array_equal(replay["actions"], z[f"{key}__actions"][:len(replay["actions"])])
    assert np.array_equal(replay["return"], z[f"{key}__return"])
