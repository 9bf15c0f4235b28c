"""Parity bar of the float32 spec (= what the HIP kernels compute) against float64 results (the
reference's golden vectors, tier 0, or the float64 oracle, tier 2), teacher-forced per step.

    masks (done, violation, remaining, scenario flags / loads)      : bit-exact
    x, y, dist_travelled, rewards, observations                     : <= 1e-5 (north_star), every scenario
    headings                                                        : <= a hand-committed constant per fixture
                                                                      class (THETA_BOUNDS below; north_star's 1e-5
                                                                      is NOT met by theta: see theta_bound)
    observation rows over 1e-5                                      : allowed ONLY when explained by a near-tie:
        the row must be built from the same own-observation blocks, and the decision that differs
        (neighbour order / membership, nearest prey, in-sensing-range) must hinge on two float64
        distances closer than TIE_BAND.  Zero unexplained rows.

`python tests/parity.py --write` regenerates tests/golden/PARITY_REPORT.json from the float32 oracle
(the HIP kernels are bit-identical to it: tests/test_gpu_parity.py::test_step_bit_exact_vs_f32_oracle).  The report is a
RECORD of the measured maxima; no limit is read back from it -- regenerating it cannot change what passes.
"""
import json
import os

import numpy as np

TOL = 1e-5             # poses, distances, rewards, observations
# Heading bounds, committed by hand per fixture class (measured maxima at the time of writing in brackets;
# PARITY_REPORT.json).  theta exceeds north_star's 1e-5 in 12 of the 45 fixtures (max 1.1e-4): a unicycle reversing
# towards a goal behind it amplifies one ulp of heading ~2.5x per controller period;
# observations never contain theta, and x, y keep the 1e-5 bar everywhere.
THETA_BOUNDS = (
    ("barrier_unsafe", 1.5e-4),   # hand-placed: two robots inside each other's safety radius, 1e6 gain  [1.08e-4]
    ("long", 5e-5),               # 74 sub-steps per step, or `robotarium: True` (a controller every sub-step)  [4.0e-5]
    ("default", 4e-5),            # 29 / 36 sub-steps, controller every 15th  [3.1e-5]
)
# WHERE the heading error comes from (round 5; the record is in PARITY_REPORT.json, per fixture).  The CONTROL is float64
# arithmetic (oracle tier 2) started from the float32-ROUNDED pre-state -- the best any engine that stores its state in float32
# can do.  `theta_control` = control vs reference: what rounding the stored state alone costs; `theta_vs_control` = float32 spec
# vs control, same rounded input: what the float32 arithmetic inside the step adds.  Measured: the control itself misses 1e-5 in
# 16 of 77 fixtures (all six barrier_unsafe fixtures: 1.1e-4 = the whole error there); the arithmetic's own share exceeds 1e-5
# in 8 (max 3.9e-5) and no single part of the step carries it -- heading in binary64, exact sin / cos, binary64 position
# controller and si -> uni map TOGETHER only bring that maximum to 2.0e-5 (NOTEBOOK.md round 5): a unicycle reversing towards
# its goal amplifies every 1e-7 of any float32 operation ~2.5x per controller period.  north_star's 1e-5 on headings is therefore
# out of reach for a float32-state engine whatever its arithmetic; x, y and the observations (no heading in them) keep it.
# Bounds on the arithmetic's own share, by class (committed by hand; measured maxima in brackets):
THETA_VS_CONTROL_BOUNDS = (
    ("barrier_unsafe", 1e-5),     # [6.8e-6]  the 1.1e-4 of this class is input rounding alone
    ("long", 5e-5),               # [3.9e-5]
    ("default", 2.5e-5),          # [2.0e-5]
)
# Two float64 distances closer than this may be ordered either way by the float32 spec: positions agree
# within ~1.2e-6 (PARITY_REPORT.json), a distance moves by at most |dp_a| + |dp_b| <= 2 sqrt(2) x that.
# The ties seen in the fixtures are exact float64 ties (robots on the 0.3 m reset grid).
TIE_BAND = 4e-6
REPORT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "PARITY_REPORT.json")
MASK_KEYS = ("prey_sensed", "prey_captured", "loaded", "load", "zone_load", "messages", "pixel_type", "reached_goal")


def angle_diff(a, b):
    return np.abs(np.angle(np.exp(1j * (np.asarray(a, np.float64) - np.asarray(b, np.float64)))))


def _layout(scenario, cfg):
    """(own-block length, neighbour slots, all_others, has_prey_fields) of an observation row."""
    if scenario == "PredatorCapturePrey":
        N = int(cfg["predator"]) + int(cfg["capture"])
        od = 6 if cfg.get("capability_aware") else 4
        K = int(cfg["num_neighbors"])
        return od, min(K, N - 1), K >= N - 1, True
    if scenario == "Warehouse":
        N = int(cfg["n_agents"])
        K = int(cfg["num_neighbors"])
        return 3, min(K, N - 1), K >= N - 1, False
    return None


def explain_row(scenario, cfg, a, got_obs_t, want_obs_t, poses_t, prey_loc_t=None, prey_captured_t=None):
    """Why observation row `a` of one step may differ from the float64 row by more than TOL.
    got_obs_t / want_obs_t: [N, D]; poses_t: float64 [3, N] post-step; prey_*: post-step.
    Returns (None, gap) when every difference is a near-tie (gap = the widest tie used), else (reason, None)."""
    lay = _layout(scenario, cfg)
    if lay is None:
        return f"{scenario} observations have no order-dependent part", None
    od, K, all_others, has_prey = lay
    N = poses_t.shape[1]
    got, want = got_obs_t[a], want_obs_t[a]
    gap = 0.0
    x, y = poses_t[0], poses_t[1]
    if np.abs(got[:2] - want[:2]).max() > TOL:
        return "own position differs", None
    if has_prey:
        npred = int(cfg["predator"])
        sr = float(cfg["predator_radius"]) if a < npred else 0.0
        if od == 6 and np.abs(got[4:6] - want[4:6]).max() > TOL:
            return "capability fields differ", None
        if np.abs(got[2:4] - want[2:4]).max() > TOL:
            d = np.hypot(prey_loc_t[:, 0] - x[a], prey_loc_t[:, 1] - y[a])
            free = ~prey_captured_t.astype(bool)
            surely = free & (d <= sr - TIE_BAND)
            maybe = free & (d <= sr + TIE_BAND)
            best = d[surely].min() if surely.any() else np.inf
            ok = False
            if not surely.any() and np.abs(got[2:4] - (-5.0)).max() <= TOL:
                ok = True                                           # in-range decision on the radius
                gap = max(gap, float(np.abs(d[maybe] - sr).max()) if maybe.any() else 0.0)
            for i in np.nonzero(maybe)[0]:
                if np.abs(got[2:4] - prey_loc_t[i]).max() <= TOL and d[i] <= best + TIE_BAND:
                    ok = True
                    gap = max(gap, float(abs(d[i] - best)) if np.isfinite(best) else float(abs(d[i] - sr)))
            if not ok:
                return "nearest-prey fields differ and no near-tie explains it", None
    # neighbour slots: each must be the own block of one other agent, as this same step computed it
    d = np.hypot(x - x[a], y - y[a])
    order = []
    for m in range(K):
        blk = got[(m + 1) * od:(m + 2) * od]
        who = [j for j in range(N) if j != a and np.array_equal(blk, got_obs_t[j, :od])]
        if len(who) != 1:
            return f"neighbour slot {m} is not the own block of exactly one other agent", None
        order.append(who[0])
    if len(set(order)) != len(order):
        return "a neighbour appears twice", None
    if all_others:
        if order != [j for j in range(N) if j != a]:
            return "all-others branch is not in index order", None
        return None, gap
    for m in range(K - 1):                                          # ascending within the band
        if d[order[m]] > d[order[m + 1]] + TIE_BAND:
            return f"neighbours {m},{m + 1} are out of order by {d[order[m]] - d[order[m + 1]]:.3g}", None
        if d[order[m]] > d[order[m + 1]]:
            gap = max(gap, float(d[order[m]] - d[order[m + 1]]))
    rest = [j for j in range(N) if j != a and j not in order]
    if rest and K:
        worst = min(d[j] for j in rest)
        if worst < d[order[-1]] - TIE_BAND:
            return f"a closer agent was left out by {d[order[-1]] - worst:.3g}", None
        if worst < d[order[-1]]:
            gap = max(gap, float(d[order[-1]] - worst))
    return None, gap


def load_report():
    with open(REPORT_PATH) as f:
        return json.load(f)


def theta_class(name, cfg):
    if "barrier_unsafe" in name:
        return "barrier_unsafe"
    if int(cfg["update_frequency"]) >= 60 or bool(cfg.get("robotarium", False)):
        return "long"
    return "default"


def theta_bound(name, cfg):
    """The committed heading bound of a fixture (by its class; nothing measured enters)."""
    return dict(THETA_BOUNDS)[theta_class(name, cfg)]


def control_run(c_oracle, scenario, cfg, state, actions):
    """The control of the heading attribution: the float64 tier stepped from the float32-rounded state.  -> OracleVecEnv."""
    from helpers import oracle_from_state
    rounded = {k: (np.asarray(v).astype(np.float32).astype(np.float64) if np.asarray(v).dtype.kind == "f" else v) for k, v in state.items()}
    env = oracle_from_state(c_oracle, scenario, cfg, rounded, np.float64)
    env.step(actions)
    return env


def theta_attribution(spec_poses, control_poses, ref_poses):
    """-> dict(theta_control, theta_vs_control, steps_over_1e5 = [spec vs ref, control vs ref, spec vs control])."""
    a, b = angle_diff(spec_poses[:, 2], ref_poses[:, 2]), angle_diff(control_poses[:, 2], ref_poses[:, 2])
    ab = angle_diff(spec_poses[:, 2], control_poses[:, 2])
    return {"theta_control": float(b.max()), "theta_vs_control": float(ab.max()),
            "steps_over_1e5": [int((a.max(axis=1) > TOL).sum()), int((b.max(axis=1) > TOL).sum()), int((ab.max(axis=1) > TOL).sum())]}


def check_step_parity(scenario, cfg, name, got, want, theta_limit=None):
    """got / want: dicts with obs [T,N,D], reward [T,N], done, viol, remaining [T], dist [T,N], poses [T,3,N]
    (post-step) and the scenario's post-step state arrays; want additionally prey_loc / prey_captured for
    PredatorCapturePrey.  Asserts the bar in this module's docstring; returns the measured maxima."""
    assert np.array_equal(got["viol"], want["viol"]), "violation codes differ"
    assert np.array_equal(got["done"], want["done"]), "done masks differ"
    assert np.array_equal(got["remaining"], want["remaining"]), "info['remaining'] differs"
    for k in MASK_KEYS:
        if k in want and want[k] is not None and k in got:
            assert np.array_equal(np.asarray(got[k]).reshape(np.asarray(want[k]).shape), want[k]), k
    m = {"rows": int(len(got["done"])),
         "max_xy": float(np.abs(got["poses"][:, :2] - want["poses"][:, :2]).max()),
         "max_theta": float(angle_diff(got["poses"][:, 2], want["poses"][:, 2]).max()),
         "max_dist": float(np.abs(got["dist"] - want["dist"]).max()),
         "max_reward": float(np.abs(got["reward"] - want["reward"]).max())}
    assert m["max_xy"] <= TOL, f"x, y off by {m['max_xy']:.3g}"
    assert m["max_dist"] <= TOL, f"dist_travelled off by {m['max_dist']:.3g}"
    assert m["max_reward"] <= TOL, f"reward off by {m['max_reward']:.3g}"
    if theta_limit is not None:
        assert m["max_theta"] <= theta_limit, f"heading off by {m['max_theta']:.3g} > {theta_limit:.3g}"
    dob = np.abs(got["obs"] - want["obs"]).max(axis=2)
    over = dob > TOL
    m["max_obs"] = float(dob[~over].max()) if (~over).any() else 0.0
    ties, widest, unexplained = 0, 0.0, []
    for t, a in zip(*np.nonzero(over)):
        why, gap = explain_row(scenario, cfg, int(a), got["obs"][t], want["obs"][t], want["poses"][t],
                               want.get("prey_loc")[t] if want.get("prey_loc") is not None else None,
                               want.get("prey_captured")[t] if want.get("prey_captured") is not None else None)
        if why is None:
            ties += 1
            widest = max(widest, gap)
        else:
            unexplained.append((int(t), int(a), why))
    assert not unexplained, f"{len(unexplained)} observation rows differ by more than {TOL} without a near-tie: {unexplained[:5]}"
    m["tie_rows"] = ties
    m["widest_tie"] = widest
    return m


def golden_want(g):
    """The reference's vectors of one fixture in check_step_parity's shape."""
    w = {"obs": g["obs"], "reward": g["reward"], "done": g["done"], "viol": g["viol"], "remaining": g["remaining"],
         "dist": g["dist"], "poses": g["post_poses"]}
    for k in MASK_KEYS:
        if "post_" + k in g.files:
            w[k] = g["post_" + k]
    if "post_prey_loc" in g.files and "post_prey_captured" in g.files:
        w["prey_loc"] = g["post_prey_loc"]
        w["prey_captured"] = g["post_prey_captured"]
    return w


def oracle_got(env):
    """An OracleVecEnv after step() in check_step_parity's shape."""
    d = {"obs": env.obs, "reward": env.reward, "done": env.done, "viol": env.viol, "remaining": env.remaining,
         "dist": env.dist, "poses": env.poses}
    for k in MASK_KEYS:
        d[k] = getattr(env, k)
    d["prey_loc"] = env.prey_loc
    d["prey_captured"] = env.prey_captured
    return d


def write_report():
    """float32 oracle vs the reference's golden vectors, every fixture -> PARITY_REPORT.json."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    from helpers import golden_files, load_golden, oracle_from_state, pre_state
    from oracle import c_oracle
    c_oracle.build_library()
    out = {"what": "float32 spec (oracle tier 3; the HIP kernels are bit-identical to it) against the reference's "
                   "golden vectors (float64, tier 0), teacher-forced per step: measured maxima per fixture",
           "tolerance": TOL, "theta_bounds": dict(THETA_BOUNDS), "theta_vs_control_bounds": dict(THETA_VS_CONTROL_BOUNDS), "tie_band": TIE_BAND,
           "generated_by": "python tests/parity.py --write", "fixtures": {}}
    for path in golden_files():
        g, scenario, cfg = load_golden(path)
        name = os.path.basename(path)[:-4]
        env = oracle_from_state(c_oracle, scenario, cfg, pre_state(g), np.float32)
        env.step(g["actions"])
        assert int(env.qp_sweeps.max()) < c_oracle.QP_MAX_SWEEPS["float32"], "a float32 QP hit its sweep cap"
        m = check_step_parity(scenario, cfg, name, oracle_got(env), golden_want(g))
        m["theta_class"] = theta_class(name, cfg)
        m["theta_bound"] = theta_bound(name, cfg)
        assert m["max_theta"] <= m["theta_bound"], (name, m["max_theta"])
        m["max_qp_sweeps"] = int(env.qp_sweeps.max())
        m["update_frequency"] = int(cfg["update_frequency"])
        m.update(theta_attribution(env.poses, control_run(c_oracle, scenario, cfg, pre_state(g), g["actions"]).poses, g["post_poses"]))
        assert m["theta_vs_control"] <= dict(THETA_VS_CONTROL_BOUNDS)[m["theta_class"]], (name, m["theta_vs_control"])
        out["fixtures"][name] = m
    fx = out["fixtures"].values()
    out["theta_summary"] = {
        "fixtures": len(out["fixtures"]),
        "spec_vs_reference_over_1e5": sum(1 for v in fx if v["max_theta"] > TOL),
        "control_vs_reference_over_1e5": sum(1 for v in fx if v["theta_control"] > TOL),
        "spec_vs_control_over_1e5": sum(1 for v in fx if v["theta_vs_control"] > TOL),
        "input_rounding_alone": sorted(k for k, v in out["fixtures"].items() if v["max_theta"] > TOL and v["theta_control"] >= 0.9 * v["max_theta"]),
        "arithmetic_dominated": sorted(k for k, v in out["fixtures"].items() if v["max_theta"] > TOL and v["theta_control"] < 0.3 * v["max_theta"]),
        "what": "control = float64 arithmetic from the float32-rounded pre-state (parity.control_run); see the comment above THETA_VS_CONTROL_BOUNDS"}
    with open(REPORT_PATH, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    return out


if __name__ == "__main__":
    import sys
    if "--write" in sys.argv:
        rep = write_report()
        worst = {k: max(v[k] for v in rep["fixtures"].values()) for k in ("max_xy", "max_theta", "max_dist", "max_reward", "max_obs")}
        print(json.dumps(worst), "tie rows:", sum(v["tie_rows"] for v in rep["fixtures"].values()))
