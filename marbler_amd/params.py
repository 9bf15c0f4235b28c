"""Scenario config (the reference's YAML keys, verbatim) -> rg_scenario_params.

Mirrors what the reference does with `objectview(yaml)` in wrapper.py:27-31 and the scenario
constructors (PredatorCapturePrey.py:15-59, warehouse.py:48-82, MaterialTransport.py:49-92),
plus the rps constants of SURVEY.md Appendix A and the reset geometry of misc.py:49-63.
All derived integers (grid sizes) are computed in float64 exactly as the reference does.
"""
import ctypes
import math
import os

import numpy as np
import yaml

from ._lib import MAX_AGENTS, MAX_PREY, RgGrid, RgScenarioParams

SCENARIO_IDS = {"PredatorCapturePrey": 0, "Warehouse": 1, "MaterialTransport": 2, "Simple": 3, "ArcticTransport": 4}
COLLISION_VARIANTS = {"center": 0, "offset": 1}
BARRIER_SOLVERS = {"exact": 0, "cvxopt": 1}      # RG_QP_EXACT, RG_QP_CVXOPT
CVXOPT_MAX_AGENTS = 8
CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")

# sim_spec_v0: which of the two upstream collision tests rps' _validate uses (SURVEY.md
# Appendix A.4).  'offset' = centres shifted 0.025 m along the heading, distance <= 0.135 m:
# the variant that ships with the per-robot dict counters the reference reads
# (roboEnv.py:84-86).  Override with config key `collision_variant: center`.
DEFAULT_COLLISION_VARIANT = "offset"

# sim_spec_v0 barrier-QP solver constants (float32): relative tolerance and sweep cap of the
# Hildreth iteration that stands in for cvxopt's interior-point qp (which the reference runs at
# reltol 1e-2).  Config keys `qp_rtol`, `qp_max_sweeps` override.
QP_RTOL = 1.25e-6
QP_MAX_SWEEPS = 40


def default_config_path(scenario):
    return os.path.join(CONFIG_DIR, scenario + ".yaml")


def load_config(scenario, config_path=None, overrides=None):
    with open(config_path or default_config_path(scenario), "r") as f:
        cfg = yaml.safe_load(f)
    if overrides:
        cfg.update(overrides)
    return cfg


def _grid(count, width, height, spacing, what):
    """rps generate_initial_conditions (Appendix A.7) geometry."""
    nx = int(np.floor(width / spacing))
    ny = int(np.floor(height / spacing))
    if nx == 0 or ny == 0:
        raise ValueError(f"{what}: spacing {spacing} too large for a {width} x {height} area")
    if not nx * ny > count:
        raise ValueError(f"{what}: {count} items need more than {nx}x{ny} = {nx * ny} grid cells "
                         f"(rps asserts cells > N; lower the spacing)")
    if nx * ny > 64:
        raise ValueError(f"{what}: {nx}x{ny} grid exceeds the 64 cells the device sampler supports")
    g = RgGrid()
    g.nx, g.ny, g.spacing, g.w2, g.h2 = nx, ny, spacing, width / 2, height / 2
    return g


def _dummy_grid():
    g = RgGrid()
    g.nx = g.ny = 1
    return g


def _check_agents(N):
    if not 1 <= N <= MAX_AGENTS:
        raise ValueError(f"n_agents must be in 1..{MAX_AGENTS}")


def make_params(scenario, cfg):
    if scenario not in SCENARIO_IDS:
        raise KeyError(f"scenario {scenario!r} is not built (have {sorted(SCENARIO_IDS)})")
    p = RgScenarioParams()
    p.scenario = SCENARIO_IDS[scenario]
    p.update_frequency = int(cfg["update_frequency"])
    # roboEnv.py:63 `if iterations % 15 == 0 or self.args.robotarium`: with `robotarium: True` (the setting of a
    # Robotarium submission) the controller runs on every sub-iteration; in simulation that is all the flag changes
    p.controller_period = 1 if cfg.get("robotarium", False) else 15
    p.max_episode_steps = int(cfg["max_episode_steps"])
    p.penalize_violations = int(bool(cfg["penalize_violations"]))
    p.shared_reward = int(bool(cfg.get("shared_reward", scenario != "Warehouse")))
    if cfg.get("real_time", False):
        raise ValueError("real_time: True paces the simulator to wall-clock time (rps sim_in_real_time): not a batched mode "
                         "(the single-env `Wrapper` honours it: marbler_amd/wrapper.py)")
    bc = cfg.get("barrier_certificate", "safe")                # roboEnv.py:15-18
    if bc == "custom" or callable(bc):
        raise ValueError("barrier_certificate: custom -- Controller(type='custom', custom=<closure>) (utilities/controller.py:17-18) hands "
                         "the QP to a Python callable, which a batched device engine cannot run.  The PARAMETRIC family rps offers is "
                         "reachable from the config instead: barrier_certificate: safe|default (certificate2 | certificate) with the "
                         "keys safety_radius, barrier_gain, unsafe_barrier_gain, magnitude_limit overriding their defaults")
    if bc not in ("safe", "default"):
        raise ValueError("barrier_certificate must be 'safe' or 'default'")
    p.barrier_has_unsafe_gain = 1 if bc == "safe" else 0       # controller.py:13-16
    # the arguments of rps' create_single_integrator_barrier_certificate{2,}: the reference passes safety_radius=.2 for 'safe'
    # and leaves the rest at rps' defaults; a config may set any of them -- what the reference would
    # write as Controller('custom', create_single_integrator_barrier_certificate2(barrier_gain=..., safety_radius=...))
    p.safety_radius = float(cfg.get("safety_radius", 0.2 if bc == "safe" else 0.17))
    p.barrier_gain = float(cfg.get("barrier_gain", 100.0))
    p.unsafe_barrier_gain = float(cfg.get("unsafe_barrier_gain", 1e6))
    p.barrier_magnitude_limit = float(cfg.get("magnitude_limit", 0.2))
    for key, val in (("safety_radius", p.safety_radius), ("barrier_gain", p.barrier_gain), ("unsafe_barrier_gain", p.unsafe_barrier_gain),
                     ("magnitude_limit", p.barrier_magnitude_limit)):
        if not (val > 0.0 and math.isfinite(val)):
            raise ValueError(f"{key} must be a positive finite number (got {val!r})")
    # How the certificate's QP is evaluated (include/robogym.h RG_QP_*).  `exact`: the projection (sim_spec_v0's default, Hildreth
    # sweeps).  `cvxopt`: the interior-point iterate the reference's stack computes -- rps hands the QP to cvxopt at reltol =
    # feastol = 1e-2, maxiters 50 (utilities/controller.py:13-16,23) -- restated, unpinned against the real
    # package; cvxopt_* keys = `cvxopt.solvers.options`.
    solver = cfg.get("barrier_solver", "exact")
    if solver not in BARRIER_SOLVERS:
        raise ValueError(f"barrier_solver must be one of {sorted(BARRIER_SOLVERS)} (got {solver!r})")
    p.qp_mode = BARRIER_SOLVERS[solver]
    p.ipm_abstol, p.ipm_reltol = float(cfg.get("cvxopt_abstol", 1e-7)), float(cfg.get("cvxopt_reltol", 1e-2))
    p.ipm_feastol, p.ipm_maxiters = float(cfg.get("cvxopt_feastol", 1e-2)), int(cfg.get("cvxopt_maxiters", 50))
    p.qp_rtol = float(cfg.get("qp_rtol", QP_RTOL))
    p.qp_max_sweeps = int(cfg.get("qp_max_sweeps", QP_MAX_SWEEPS))
    p.collision_variant = COLLISION_VARIANTS[cfg.get("collision_variant", DEFAULT_COLLISION_VARIANT)]
    p.time_step = 0.033
    p.bound_x0, p.bound_y0, p.bound_w, p.bound_h = -1.6, -1.0, 3.2, 2.0
    p.robot_diameter, p.wheel_radius, p.max_linear_velocity = 0.11, 0.016, 0.2
    p.collision_offset, p.collision_diameter = 0.025, 0.135
    p.projection_distance, p.angular_velocity_limit, p.position_velocity_limit = 0.05, math.pi, 0.15
    L, R, U, D = cfg["LEFT"], cfg["RIGHT"], cfg["UP"], cfg["DOWN"]
    p.left, p.right, p.up, p.down = L, R, U, D
    height = D - U
    if scenario == "PredatorCapturePrey":
        npred, ncap = int(cfg["predator"]), int(cfg["capture"])
        N = npred + ncap
        p.n_agents = N
        _check_agents(N)
        p.capability_aware = int(bool(cfg["capability_aware"]))
        p.num_prey = int(cfg["num_prey"])
        if not 1 <= p.num_prey <= MAX_PREY:
            raise ValueError(f"num_prey must be in 1..{MAX_PREY}")
        p.num_neighbors = int(cfg["num_neighbors"])
        p.obs_dim = (6 if p.capability_aware else 4) * (p.num_neighbors + 1)
        for a in range(N):
            p.agent_step[a] = cfg["step_dist"]
            p.sensing_radius[a] = cfg["predator_radius"] if a < npred else 0.0
            p.capture_radius[a] = 0.0 if a < npred else cfg["capture_radius"]
        p.time_penalty, p.sense_reward, p.capture_reward = \
            cfg["time_penalty"], cfg["sense_reward"], cfg["capture_reward"]
        p.violation_reward = -5.0                               # PredatorCapturePrey.py:159
        thresh = cfg["ROBOT_INIT_RIGHT_THRESH"]
        width = thresh - L                                     # PredatorCapturePrey.py:123-125
        g = _grid(N, width, height, cfg["start_dist"], "agents")
        g.ox1 = -(width / 2 - thresh)                           # misc.py:57
        p.agent_grid = g
        width = R - cfg["PREY_INIT_LEFT_THRESH"]               # :128-129 (shift uses ROBOT_INIT_RIGHT_THRESH)
        q = _grid(p.num_prey, width, height, cfg["step_dist"], "prey")
        q.ox1 = (width / 2 - thresh)                            # misc.py:61
        p.prey_grid = q
        p.keep_theta = 0
    elif scenario == "Warehouse":
        N = int(cfg["n_agents"])
        p.n_agents = N
        _check_agents(N)
        p.num_neighbors = int(cfg["num_neighbors"])
        p.obs_dim = 3 * (p.num_neighbors + 1)
        for a in range(N):
            p.agent_step[a] = cfg["step_dist"]
        p.load_reward, p.unload_reward, p.goal_width = cfg["load_reward"], cfg["unload_reward"], cfg["goal_width"]
        p.violation_reward = -5.0                               # warehouse.py:116
        g = _grid(N, R - L, height, cfg["start_dist"], "agents")   # warehouse.py:90-98
        g.ox1, g.ox2 = (1.5 + L) / 2, -((1.5 - R) / 2)
        g.oy1, g.oy2 = -((1 + U) / 2), (1 - D) / 2
        p.agent_grid = g
        p.prey_grid = _dummy_grid()
        p.keep_theta = 1
    elif scenario == "Simple":
        N = int(cfg["n_agents"])
        p.n_agents = N
        _check_agents(N)
        p.num_prey = 1                                          # the single goal (simple.py:74)
        p.obs_dim = 2 * (N + 1)                                 # simple.py:98
        for a in range(N):
            p.agent_step[a] = cfg["step_dist"]
        p.reward_scaler = cfg["reward_scaler"]
        p.violation_reward = -5.0                               # simple.py:174
        thresh = cfg["ROBOT_INIT_RIGHT_THRESH"]
        width = thresh - L                                     # simple.py:136-140
        g = _grid(N, width, height, cfg["start_dist"], "agents")
        g.ox1 = -(width / 2 - thresh)
        p.agent_grid = g
        width = R - cfg["PREY_INIT_LEFT_THRESH"]               # simple.py:143-146
        q = _grid(1, width, height, cfg["step_dist"], "goal")
        q.ox1 = (width / 2 - thresh)
        p.prey_grid = q
        p.keep_theta = 0
    elif scenario == "ArcticTransport":
        N = int(cfg["n_agents"])
        if N != 4:
            raise ValueError("ArcticTransport is hard-wired to 4 agents (ArcticTransport.py:26-31)")
        p.n_agents = 4
        p.obs_dim = 30                                          # ArcticTransport.py:19
        p.arctic_normal_step, p.arctic_slow_step, p.arctic_fast_step = \
            cfg["normal_step"], cfg["slow_step"], cfg["fast_step"]
        p.not_reached_penalty, p.dist_multiplier = cfg["not_reached_penalty"], cfg["dist_multiplier"]
        p.violation_reward = -30.0                              # ArcticTransport.py:103
        g = _dummy_grid()                                       # fixed start poses (ArcticTransport.py:30-33)
        g.nx, g.ny = 8, 1
        p.agent_grid = g
        p.prey_grid = _dummy_grid()
        p.keep_theta = 0
    else:
        nf, ns = int(cfg["n_fast_agents"]), int(cfg["n_slow_agents"])
        N = int(cfg["n_agents"])
        if nf + ns != N:
            raise ValueError("n_fast_agents + n_slow_agents must equal n_agents")
        if N < 4:
            raise ValueError("MaterialTransport broadcasts the messages of agents 0-3: n_agents >= 4")
        p.n_agents = N
        _check_agents(N)
        p.capability_aware = int(bool(cfg["capability_aware"]))
        p.obs_dim = 11 if p.capability_aware else 9
        for a in range(N):
            p.agent_step[a] = cfg["fast_step"] if a < nf else cfg["slow_step"]
            p.torque[a] = int(cfg["small_torque"] if a < nf else cfg["large_torque"])
        p.time_penalty = cfg["time_penalty"]
        p.unload_multiplier, p.load_multiplier = cfg["unload_multiplier"], cfg["load_multiplier"]
        p.end_goal_width, p.zone1_radius = cfg["end_goal_width"], cfg["zone1_radius"]
        p.violation_reward = -6.0                               # MaterialTransport.py:137
        for z in ("zone1", "zone2"):
            if cfg[z].get("distribution", "normal") != "normal":
                raise ValueError("only the 'normal' zone distribution of the reference config is built")
        p.zone1_mean, p.zone1_std = cfg["zone1"]["loc"], cfg["zone1"]["scale"]
        p.zone2_mean, p.zone2_std = cfg["zone2"]["loc"], cfg["zone2"]["scale"]
        width = cfg["end_goal_width"]                           # MaterialTransport.py:106-109
        thresh = L + cfg["end_goal_width"]
        g = _grid(N, width, height, cfg["start_dist"], "agents")
        g.ox1 = -(width / 2 - thresh)
        p.agent_grid = g
        p.prey_grid = _dummy_grid()
        p.keep_theta = 0
    if p.qp_mode == BARRIER_SOLVERS["cvxopt"] and p.n_agents > CVXOPT_MAX_AGENTS:
        raise ValueError(f"barrier_solver: cvxopt is built for n_agents <= {CVXOPT_MAX_AGENTS} (its QP is solved in one lane: 2N unknowns, "
                         f"N(N-1)/2 rows in registers); got {p.n_agents}")
    return p


def params_to_bytes(p):
    return bytes(ctypes.string_at(ctypes.addressof(p), ctypes.sizeof(p)))


def params_from_bytes(b):
    p = RgScenarioParams()
    ctypes.memmove(ctypes.addressof(p), bytes(b), ctypes.sizeof(p))
    return p


if __name__ == "__main__":   # python -m marbler_amd.params <Scenario> <out.bin> [key=value ...]: the YAML as an rg_scenario_params blob
    import sys
    if len(sys.argv) < 3:
        raise SystemExit("usage: python -m marbler_amd.params <Scenario> <out.bin> [key=value ...]")
    ov = {}
    for kv in sys.argv[3:]:
        k, v = kv.split("=", 1)
        ov[k] = yaml.safe_load(v)
    if sys.argv[1] == "PredatorCapturePrey" and ("predator" in ov or "capture" in ov):
        cfg0 = load_config(sys.argv[1])
        ov.setdefault("n_agents", int(ov.get("predator", cfg0["predator"])) + int(ov.get("capture", cfg0["capture"])))
    with open(sys.argv[2], "wb") as fh:
        fh.write(params_to_bytes(make_params(sys.argv[1], load_config(sys.argv[1], overrides=ov))))
    print(f"wrote {sys.argv[2]} ({ctypes.sizeof(RgScenarioParams)} bytes)")
