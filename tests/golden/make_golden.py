"""Generate tests/golden/*.npz from the reference's own scenario code (run HERE only).

    python tests/golden/make_golden.py

Every fixture is a sequence of single env steps recorded teacher-forced: the full state
before the step, the actions, everything `step()` returned, and the state after.  The
simulator below the reference's layers is the restated rps (oracle/rps_restated), so these
vectors pin rows a1, a2, a11-a16 of SURVEY.md section 8 bit-for-bit in float64 GIVEN that
simulator; rows a4-a10 stay "parity unpinned" against real rps + cvxopt.

Environment recorded with the vectors: numpy version (the neighbour order produced by
np.argpartition is implementation-defined).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402


# ----------------------------------------------------------------------------- policies
def _toward(dx, dy):
    if abs(dx) >= abs(dy):
        return 1 if dx > 0 else 0
    return 3 if dy > 0 else 2


def policy_pcp(w, rng, eps):
    s = w.env
    acts = []
    for a, ag in enumerate(s.agents):
        if rng.rand() < eps:
            acts.append(int(rng.randint(5)))
            continue
        p = s.agent_poses[:2, a]
        best, bd = None, 1e9
        for i in range(len(s.prey_loc)):
            if s.prey_captured[i]:
                continue
            d = np.linalg.norm(p - s.prey_loc[i])
            if d < bd:
                best, bd = i, d
        if best is None:
            acts.append(4)
        elif ag.capture_radius > 0 and bd <= ag.capture_radius * 0.9:
            acts.append(4)
        else:
            dx, dy = s.prey_loc[best] - p
            acts.append(_toward(dx, dy))
    return acts


def policy_warehouse(w, rng, eps):
    s = w.env
    acts = []
    for a, ag in enumerate(s.agents):
        if rng.rand() < eps:
            acts.append(int(rng.randint(5)))
            continue
        p = s.agent_poses[:2, a]
        if ag.loaded:
            ty = 0.4 if ag.goal == 'Green' else -0.4
            tx = -1.4
        else:
            ty = 0.4 if ag.goal == 'Red' else -0.4
            tx = 1.4
        if abs(p[1] - ty) > 0.25 and rng.rand() < 0.5:
            acts.append(3 if ty > p[1] else 2)
        else:
            acts.append(1 if tx > p[0] else 0)
    return acts


def policy_mt(w, rng, eps):
    s = w.env
    acts = []
    for a, ag in enumerate(s.agents):
        msg = int(rng.randint(4))
        if rng.rand() < eps:
            acts.append(int(rng.randint(20)))
            continue
        p = s.agent_poses[:2, a]
        if ag.load > 0:
            mv = 0
        elif a % 2 == 0 and s.zone1_load > 0:
            dx, dy = -p[0], -p[1]
            mv = 4 if np.hypot(dx, dy) < 0.2 else _toward(dx, dy)
        elif s.zone2_load > 0:
            mv = 1
        elif s.zone1_load > 0:
            dx, dy = -p[0], -p[1]
            mv = 4 if np.hypot(dx, dy) < 0.2 else _toward(dx, dy)
        else:
            mv = 4
        acts.append(mv * 4 + msg)
    return acts


def policy_simple(w, rng, eps):
    s = w.env
    acts = []
    for a in range(s.num_robots):
        if rng.rand() < eps:
            acts.append(int(rng.randint(5)))
            continue
        dx, dy = np.asarray(s.goal_loc).reshape(-1) - s.agent_poses[:2, a]
        acts.append(4 if np.hypot(dx, dy) < 0.15 else _toward(dx, dy))
    return acts


def policy_arctic(w, rng, eps):
    s = w.env
    acts = []
    gx, gy = s.get_pose_from_cell(s.goal_loc)
    for a in range(s.num_robots):
        if rng.rand() < eps:
            acts.append(int(rng.randint(5)))
            continue
        if a < 2:   # drones wander ahead of the ground robots
            tx, ty = gx + (0.4 if a else -0.4), 0.2
        else:
            tx, ty = gx - 0.12 + 0.25 * (a - 2) * 0.9, gy + 0.1
        dx, dy = tx - s.agent_poses[0, a], ty - s.agent_poses[1, a]
        acts.append(4 if np.hypot(dx, dy) < 0.12 else _toward(dx, dy))
    return acts


POLICY = {"PredatorCapturePrey": policy_pcp, "Warehouse": policy_warehouse, "MaterialTransport": policy_mt,
          "Simple": policy_simple, "ArcticTransport": policy_arctic}

# ----------------------------------------------------------------------------- cases
CASES = [
    # name, scenario, overrides, seeds, steps per seed, eps (prob. of a uniformly random action)
    ("pcp_n5", "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, [11, 12, 13], 140, 0.35),
    ("pcp_n4_default", "PredatorCapturePrey", {}, [21, 22], 120, 0.35),
    ("pcp_n5_random", "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, [31], 170, 1.0),
    ("pcp_n6_capaware", "PredatorCapturePrey", {"predator": 3, "capture": 3, "n_agents": 6, "capability_aware": True,
                                                "num_neighbors": 2}, [41], 100, 0.4),
    ("warehouse_n8", "Warehouse", {"n_agents": 8}, [51, 52], 150, 0.3),
    ("warehouse_n6_default", "Warehouse", {}, [61], 130, 0.3),
    ("warehouse_n8_random", "Warehouse", {"n_agents": 8}, [71], 110, 1.0),
    ("mt_n4_default", "MaterialTransport", {}, [81, 82], 100, 0.25),
    ("mt_n6", "MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25},
     [91, 92], 100, 0.25),
    ("mt_n6_random", "MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25},
     [95], 80, 1.0),
    ("mt_n4_capaware", "MaterialTransport", {"capability_aware": True}, [97], 60, 0.3),
    ("simple_n4_default", "Simple", {}, [101, 102], 110, 0.3),
    ("simple_n6_random", "Simple", {"n_agents": 6}, [105], 110, 1.0),
    ("arctic_default", "ArcticTransport", {}, [111, 112, 113], 130, 0.25),
    ("arctic_random", "ArcticTransport", {}, [115], 130, 1.0),
    # `robotarium: True`: the controller (and its QP) on every sub-iteration (roboEnv.py:63)
    ("pcp_n5_robotarium", "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "robotarium": True}, [121], 90, 0.35),
    ("mt_n4_robotarium", "MaterialTransport", {"robotarium": True}, [123], 40, 0.3),
    # the other barrier certificate (controller.py:15-16: r = 0.17, no unsafe gain), violations not penalised
    # (roboEnv.py:82: no early exit, robots may overlap), the other upstream collision test (Appendix A.4)
    ("pcp_n5_cert_default", "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "barrier_certificate": "default"},
     [131, 132], 120, 0.35),
    ("warehouse_n8_no_penalty", "Warehouse", {"n_agents": 8, "penalize_violations": False}, [141], 140, 0.5),
    ("pcp_n5_center_collision", "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "collision_variant": "center"},
     [151], 170, 1.0),
    # round 5 -- other arguments of rps' certificate factories (config keys safety_radius / barrier_gain / unsafe_barrier_gain /
    # magnitude_limit; the reference's own route: Controller('custom', create_..._certificate2(...)), controller.py:17-18)
    ("pcp_n5_family", "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "safety_radius": 0.25, "barrier_gain": 30.0,
                                              "unsafe_barrier_gain": 1e5, "magnitude_limit": 0.12}, [161], 120, 0.5),
    ("warehouse_n6_family_default", "Warehouse", {"barrier_certificate": "default", "safety_radius": 0.22, "barrier_gain": 300.0}, [163], 100, 0.5),
    # round 5 -- `barrier_solver: cvxopt`: the certificate's QP handed to the restated interior-point `qp` at rps' options
    # (reltol = feastol = 1e-2, maxiters 50), as the reference's own stack evaluates it (controller.py:13-16,23): ipm_* fixtures
    ("ipm_pcp_n5", "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "barrier_solver": "cvxopt"}, [211, 212, 213], 140, 0.35),
    ("ipm_pcp_n4_default", "PredatorCapturePrey", {"barrier_solver": "cvxopt"}, [221, 222], 120, 0.35),
    ("ipm_pcp_n5_random", "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "barrier_solver": "cvxopt"}, [231], 170, 1.0),
    ("ipm_warehouse_n8", "Warehouse", {"n_agents": 8, "barrier_solver": "cvxopt"}, [251], 150, 0.3),
    ("ipm_warehouse_n6_default", "Warehouse", {"barrier_solver": "cvxopt"}, [261], 130, 0.3),
    ("ipm_warehouse_n8_random", "Warehouse", {"n_agents": 8, "barrier_solver": "cvxopt"}, [271], 110, 1.0),
    ("ipm_mt_n4_default", "MaterialTransport", {"barrier_solver": "cvxopt"}, [281], 100, 0.25),
    ("ipm_mt_n6", "MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25, "barrier_solver": "cvxopt"},
     [291], 100, 0.25),
    ("ipm_mt_n6_random", "MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25,
                                               "barrier_solver": "cvxopt"}, [295], 80, 1.0),
    ("ipm_simple_n4_default", "Simple", {"barrier_solver": "cvxopt"}, [301], 110, 0.3),
    ("ipm_simple_n6_random", "Simple", {"n_agents": 6, "barrier_solver": "cvxopt"}, [305], 110, 1.0),
    ("ipm_arctic_default", "ArcticTransport", {"barrier_solver": "cvxopt"}, [311, 312], 130, 0.25),
    ("ipm_pcp_n5_robotarium", "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "robotarium": True, "barrier_solver": "cvxopt"},
     [321], 90, 0.35),
    ("ipm_pcp_n5_cert_default", "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "barrier_certificate": "default",
                                                        "barrier_solver": "cvxopt"}, [331], 120, 0.35),
    ("ipm_pcp_n5_family", "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "safety_radius": 0.25, "barrier_gain": 30.0,
                                                  "unsafe_barrier_gain": 1e5, "magnitude_limit": 0.12, "barrier_solver": "cvxopt"}, [341], 100, 0.5),
]


def run_case(name, scenario, overrides, seeds, steps, eps):
    recs = []
    first = []
    cfg_out = None
    for seed in seeds:
        ov = dict(overrides)
        ov["seed"] = seed
        w, cfg = rh.make_reference_wrapper(scenario, ov, collision_variant=ov.get("collision_variant", "offset"))
        cfg_out = cfg
        rng = np.random.RandomState(1000 + seed)
        rh.quiet_reset(w)
        after_reset = True
        for t in range(steps):
            acts = POLICY[scenario](w, rng, eps)
            rec = rh.step_record(w, scenario, acts)
            recs.append(rec)
            first.append(after_reset)
            after_reset = False
            if rec["done"]:
                rh.quiet_reset(w)
                after_reset = True
    return recs, np.array(first, dtype=np.uint8), cfg_out


def forced_violation_cases(solver="exact"):
    """Hand-placed poses that trip rps' validation (collision / boundary / both) on a later step;
    one clean step is taken first (roboEnv.py:21 initialises its counters to ints, so a
    violation on the very first validate of a process would raise in the reference).
    solver = "cvxopt": the same with the interior-point stand-in (ipm_viol_* fixtures, three scenarios)."""
    out = []
    scenarios = (("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}),
                 ("Warehouse", {"n_agents": 8}),
                 ("MaterialTransport", {}), ("Simple", {}), ("ArcticTransport", {}))
    prefix = ""
    if solver == "cvxopt":
        scenarios = tuple((sc, dict(ov, barrier_solver="cvxopt")) for sc, ov in scenarios[:3])
        prefix = "ipm_"
    for scenario, ov in scenarios:
        for kind in ("collision", "boundary", "both", "barrier_unsafe", "late_boundary"):
            ov2 = dict(ov)
            ov2["seed"] = 5
            if kind == "late_boundary":
                ov2["RIGHT"] = 1.75   # goal beyond the arena: the robot drives out mid-step
            w, cfg = rh.make_reference_wrapper(scenario, ov2)
            rh.quiet_reset(w)
            N = w.env.num_robots
            noop = [4] * N if scenario != "MaterialTransport" else [16] * N
            recs = [rh.step_record(w, scenario, noop)]
            P = w.env.agent_poses  # live alias of the simulator state
            acts = list(noop)
            if kind in ("collision", "both"):
                P[0, 1] = P[0, 0] + 0.09
                P[1, 1] = P[1, 0] + 0.01
                P[2, 1] = P[2, 0]
            if kind in ("boundary", "both"):
                P[0, N - 1] = 1.62
                P[1, N - 1] = 0.3
            if kind == "late_boundary":
                P[0, N - 1], P[1, N - 1], P[2, N - 1] = 1.55, 0.3, 0.0
                acts[N - 1] = 1 * 4 if scenario == "MaterialTransport" else 1
            if kind == "barrier_unsafe":
                # two robots inside each other's safety radius, commanded at each other: the
                # certificate's unsafe branch (h < 0, gain 1e6) is active
                P[0, 0], P[1, 0], P[2, 0] = 0.07, 0.06, 0.0     # off the 0.25 m ArcticTransport cell lines
                P[0, 1], P[1, 1], P[2, 1] = 0.26, 0.06, np.pi
                if scenario == "MaterialTransport":
                    acts[0], acts[1] = 1 * 4, 0 * 4
                else:
                    acts[0], acts[1] = 1, 0
            recs.append(rh.step_record(w, scenario, acts))
            out.append((f"{prefix}viol_{scenario}_{kind}", scenario, cfg, recs))
    return out


def pack(recs):
    keys = recs[0].keys()
    return {k: np.stack([np.asarray(r[k]) for r in recs]) for k in keys}


def main():
    assert rh.reference_available(), "run in the build container (needs /root/reference)"
    import json
    only = [a for a in sys.argv[1:] if not a.startswith("-")]      # python make_golden.py [fixture names]: just those
    for name, scenario, ov, seeds, steps, eps in CASES:
        if only and name not in only:
            continue
        recs, first, cfg = run_case(name, scenario, ov, seeds, steps, eps)
        d = pack(recs)
        d["first_after_reset"] = first
        d["seeds"] = np.array(seeds, dtype=np.int64)
        d["steps_per_seed"] = np.int64(steps)
        d["config_json"] = np.array(json.dumps(cfg))
        d["scenario"] = np.array(scenario)
        d["numpy_version"] = np.array(np.__version__)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        v = d["viol"]
        print(f"{name}: T={len(recs)} done={int(d['done'].sum())} viol={int((v > 0).sum())} "
              f"reward_sum={d['reward'][:, 0].sum():.3f}")
    viol = [] if only and "viol" not in only and "ipm_viol" not in only else \
        (forced_violation_cases() if (not only or "viol" in only) else []) + (forced_violation_cases("cvxopt") if (not only or "ipm_viol" in only) else [])
    for name, scenario, cfg, recs in viol:
        d = pack(recs)
        d["first_after_reset"] = np.array([1] + [0] * (len(recs) - 1), dtype=np.uint8)
        d["config_json"] = np.array(json.dumps(cfg))
        d["scenario"] = np.array(scenario)
        d["numpy_version"] = np.array(np.__version__)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(f"{name}: viol codes {d['viol'].tolist()} done {d['done'].tolist()}")


if __name__ == "__main__":
    main()
