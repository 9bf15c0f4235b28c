"""The 1e-5 bar beyond the 45 fixtures (CPU tier): the float32 spec (= the HIP kernels, bit for bit) against the float64
oracle (= the reference's arithmetic, tests/test_oracle_golden.py), TEACHER-FORCED per step from the float64 trajectory's
states (cast to float32), over ~100 000 env steps of free-running random-policy rollouts per configuration.

What this measures, and what it found (round 3; numbers in DESIGN.md section 2):
  * the fixtures sit away from threshold ties; random rollouts do not, so now and then float32 rounding flips a decision (the
    boundary crossed one sub-step earlier, a prey sensed at the radius, a neighbour order at a tie): such steps are COUNTED and
    must be rare;
  * on every other step x, y and dist_travelled agree within north_star's 1e-5 for >= 99.9 % of the env steps of every
    scenario -- but NOT for all of them: a unicycle reversing towards a goal behind it amplifies a heading error ~2.5x per
    controller period, and a pair inside the safety radius (the 1e6 gain) is stiff; over MaterialTransport's five periods per
    step about 1.5e-4 of the steps end 1e-5 .. 1e-4 apart (29-sub-step scenarios: < 2e-5 of the steps, up to 2.6e-5).  Such a
    step amplifies ANY 1e-7 perturbation a thousandfold: the rounding of the float32 state it starts from (float64 arithmetic
    from the rounded state is itself up to 3.9e-5 off) as much as the roundings of the float32 arithmetic inside it (up to
    5.9e-5 from identical input).  The bound asserted here is the measured one with margin, and it is stated wherever the
    1e-5 claim is made.
"""
import numpy as np
import pytest

from helpers import oracle_reset, oracle_reset_params

CASES = [("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5),
         ("PredatorCapturePrey", {}, 5),
         ("Warehouse", {"n_agents": 8}, 5),
         ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20),
         ("Simple", {}, 5),
         ("ArcticTransport", {}, 5)]


def measure(oracle_lib, scenario, ov, n_act, E, T, threads=4):
    from marbler_amd.params import load_config, make_params
    cfg = load_config(scenario, None, ov)
    p = make_params(scenario, cfg)
    a = oracle_lib.OracleVecEnv(scenario, cfg, E, dtype=np.float64)      # drives the trajectory
    b = oracle_lib.OracleVecEnv(scenario, cfg, E, dtype=np.float32)      # re-seeded from a's state before every step
    rp = oracle_reset_params(oracle_lib, p)
    for e in range(E):
        oracle_reset(oracle_lib, a, rp, 5, e, 0)
    episodes = np.zeros(E, np.int64)
    rng = np.random.RandomState(11)
    xy, flips, rows_over, rows, worst_obs = [], 0, 0, 0, 0.0
    for t in range(T):
        for k in a.STATE_KEYS:
            src, dst = getattr(a, k), getattr(b, k)
            dst[...] = src.astype(dst.dtype)
        act = rng.randint(0, n_act, size=(E, a.N)).astype(np.int32)
        a.step(act, threads=threads)
        b.step(act, threads=threads)
        same = (a.done == b.done) & (a.viol == b.viol) & (a.remaining == b.remaining)
        # a violation found one sub-step apart ends the step one Euler step apart: the same kind of flip
        same &= ~((a.viol > 0) & (np.abs(a.dist - b.dist).max(axis=1) > 1e-3))
        flips += int((~same).sum())
        d_xy = np.maximum(np.abs(a.poses[:, :2] - b.poses[:, :2]).max(axis=(1, 2)), np.abs(a.dist - b.dist).max(axis=1))
        xy.append(d_xy[same])
        calm = same & (d_xy <= 1e-5)
        d = np.abs(a.obs[calm] - b.obs[calm]).max(axis=2)                # [envs, agents]: per observation row
        rows += d.size
        over = d > 1e-5
        rows_over += int(over.sum())
        if (~over).any():
            worst_obs = max(worst_obs, float(d[~over].max()))
        for e in np.nonzero(a.done)[0]:
            episodes[e] += 1
            oracle_reset(oracle_lib, a, rp, 5, e, int(episodes[e]))
    xy = np.concatenate(xy)
    return {"env_steps": E * T, "decision_flips": flips, "xy_over_1e-5": int((xy > 1e-5).sum()), "xy_max": float(xy.max()),
            "xy_q999": float(np.quantile(xy, 0.999)), "obs_rows": rows, "obs_rows_over_1e-5": rows_over, "obs_max_rest": worst_obs,
            "episodes": int(episodes.sum())}


@pytest.mark.parametrize("scenario,ov,n_act", CASES)
def test_float32_spec_against_float64_from_random_states(scenario, ov, n_act, oracle_lib):
    m = measure(oracle_lib, scenario, ov, n_act, 512, 200 if scenario != "MaterialTransport" else 120)
    n = m["env_steps"]
    assert m["decision_flips"] <= max(3, 2e-4 * n), m
    assert m["xy_q999"] <= 1e-5, m                                       # north_star's bar on >= 99.9 % of the steps ...
    assert m["xy_over_1e-5"] <= max(2, 6e-4 * n) and m["xy_max"] <= 3e-4, m   # ... the rest is rare and small (docstring)
    assert m["obs_rows_over_1e-5"] <= max(3, 1e-3 * m["obs_rows"]) and m["obs_max_rest"] <= 1e-5, m
    assert m["episodes"] > 256


if __name__ == "__main__":   # python tests/test_float_spec_random.py : the measurement at a larger size, for DESIGN.md
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import c_oracle
    c_oracle.build_library()
    for scenario, ov, n_act in CASES:
        print(scenario, ov, json.dumps(measure(c_oracle, scenario, ov, n_act, 2048, 300 if scenario != "MaterialTransport" else 150, threads=8)))
