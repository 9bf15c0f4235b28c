"""ORACLE -- ctypes binding of oracle/_build/liboracle.so (test infrastructure only).

`OracleVecEnv` steps E independent envs on the CPU in float64 (tier 2: libm, follows the
reference + restated rps operation for operation) or float32 (tier 3: the spec'd float
arithmetic the HIP kernels reproduce bit for bit).  See oracle_core.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_OVERRIDE = os.environ.get("ORACLE_LIB")   # e.g. the ASan + UBSan build (`make -C oracle asan`, tests/test_sanitizers.py)
MAXN, MAXP = 16, 64
SCN = {"PredatorCapturePrey": 0, "Warehouse": 1, "MaterialTransport": 2, "Simple": 3, "ArcticTransport": 4}


class OrcParams(C.Structure):
    _fields_ = [
        ("scenario", C.c_int32), ("n_agents", C.c_int32), ("obs_dim", C.c_int32),
        ("update_frequency", C.c_int32), ("controller_period", C.c_int32),
        ("max_episode_steps", C.c_int32), ("penalize_violations", C.c_int32),
        ("barrier_has_unsafe_gain", C.c_int32), ("collision_variant", C.c_int32),
        ("capability_aware", C.c_int32), ("num_prey", C.c_int32), ("num_neighbors", C.c_int32),
        ("torque", C.c_int32 * MAXN),
        ("time_step", C.c_double), ("bound_x0", C.c_double), ("bound_y0", C.c_double),
        ("bound_w", C.c_double), ("bound_h", C.c_double),
        ("robot_diameter", C.c_double), ("wheel_radius", C.c_double), ("max_linear_velocity", C.c_double),
        ("collision_offset", C.c_double), ("collision_diameter", C.c_double),
        ("projection_distance", C.c_double), ("angular_velocity_limit", C.c_double),
        ("position_velocity_limit", C.c_double),
        ("barrier_gain", C.c_double), ("unsafe_barrier_gain", C.c_double), ("safety_radius", C.c_double),
        ("barrier_magnitude_limit", C.c_double), ("qp_rtol", C.c_double), ("qp_max_sweeps", C.c_int32),
        ("qp_mode", C.c_int32),
        ("ipm_abstol", C.c_double), ("ipm_reltol", C.c_double), ("ipm_feastol", C.c_double), ("ipm_maxiters", C.c_int32), ("pad_", C.c_int32),
        ("left", C.c_double), ("right", C.c_double), ("up", C.c_double), ("down", C.c_double),
        ("agent_step", C.c_double * MAXN), ("sensing_radius", C.c_double * MAXN),
        ("capture_radius", C.c_double * MAXN),
        ("time_penalty", C.c_double), ("sense_reward", C.c_double), ("capture_reward", C.c_double),
        ("violation_reward", C.c_double),
        ("load_reward", C.c_double), ("unload_reward", C.c_double), ("goal_width", C.c_double),
        ("unload_multiplier", C.c_double), ("load_multiplier", C.c_double), ("end_goal_width", C.c_double),
        ("zone1_radius", C.c_double), ("reward_scaler", C.c_double),
        ("arctic_normal_step", C.c_double), ("arctic_slow_step", C.c_double), ("arctic_fast_step", C.c_double),
        ("not_reached_penalty", C.c_double), ("dist_multiplier", C.c_double),
    ]


def build_library(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                              for f in ("oracle.c", "oracle_core.h", "oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_OVERRIDE) if _OVERRIDE else C.CDLL(build_library())
        assert _lib.orc_sizeof_params() == C.sizeof(OrcParams), "orc_params layout mismatch"
    return _lib


QP_RTOL = {"float32": 1.25e-6, "float64": 5e-12}
QP_MAX_SWEEPS = {"float32": 40, "float64": 200}


def params_from_config(scenario, cfg, collision_variant="offset", dtype=np.float64):
    """YAML keys of the reference's scenario config.yaml -> orc_params (Appendix A/B constants)."""
    p = OrcParams()
    p.qp_rtol = cfg.get("qp_rtol", QP_RTOL[np.dtype(dtype).name])
    p.qp_max_sweeps = cfg.get("qp_max_sweeps", QP_MAX_SWEEPS[np.dtype(dtype).name])
    # barrier_solver: exact (the projection by Hildreth sweeps) | cvxopt (the restated interior-point iterate at rps' options);
    # `qp_solver: cvxopt_restated` is the round-3/4 spelling of the latter.  ipm_spec_f64: the float spec's operation order in float64.
    solver = cfg.get("barrier_solver", "cvxopt" if cfg.get("qp_solver", "exact") == "cvxopt_restated" else "exact")
    assert solver in ("exact", "cvxopt", "ipm_spec"), solver
    p.qp_mode = {"exact": 0, "cvxopt": 1, "ipm_spec": 2}[solver]
    p.ipm_abstol, p.ipm_reltol = cfg.get("cvxopt_abstol", 1e-7), cfg.get("cvxopt_reltol", 1e-2)
    p.ipm_feastol, p.ipm_maxiters = cfg.get("cvxopt_feastol", 1e-2), int(cfg.get("cvxopt_maxiters", 50))
    p.scenario = SCN[scenario]
    p.update_frequency = int(cfg["update_frequency"])
    p.controller_period = 1 if cfg.get("robotarium", False) else 15   # roboEnv.py:63
    p.max_episode_steps = int(cfg["max_episode_steps"])
    p.penalize_violations = int(bool(cfg["penalize_violations"]))
    bc = cfg.get("barrier_certificate", "safe")   # roboEnv.py:15-18: absent -> Controller() -> 'safe'
    assert bc in ("safe", "default")
    p.barrier_has_unsafe_gain = 1 if bc == "safe" else 0
    p.safety_radius = cfg.get("safety_radius", 0.2 if bc == "safe" else 0.17)   # controller.py:13-16; the keys below: the
    p.barrier_gain = cfg.get("barrier_gain", 100.0)                            # arguments of rps' certificate factories (A.6)
    p.unsafe_barrier_gain = cfg.get("unsafe_barrier_gain", 1e6)
    p.barrier_magnitude_limit = cfg.get("magnitude_limit", 0.2)
    p.collision_variant = {"center": 0, "offset": 1}[cfg.get("collision_variant", collision_variant)]
    p.time_step = 0.033
    p.bound_x0, p.bound_y0, p.bound_w, p.bound_h = -1.6, -1.0, 3.2, 2.0
    p.robot_diameter, p.wheel_radius, p.max_linear_velocity = 0.11, 0.016, 0.2
    p.collision_offset, p.collision_diameter = 0.025, 0.135
    p.projection_distance, p.angular_velocity_limit, p.position_velocity_limit = 0.05, np.pi, 0.15
    p.left, p.right, p.up, p.down = cfg["LEFT"], cfg["RIGHT"], cfg["UP"], cfg["DOWN"]
    if scenario == "PredatorCapturePrey":
        npred, ncap = int(cfg["predator"]), int(cfg["capture"])
        N = npred + ncap
        p.n_agents = N
        p.capability_aware = int(bool(cfg["capability_aware"]))
        p.num_prey = int(cfg["num_prey"])
        p.num_neighbors = int(cfg["num_neighbors"])
        od = 6 if p.capability_aware else 4
        p.obs_dim = od * (p.num_neighbors + 1)
        for a in range(N):
            p.agent_step[a] = cfg["step_dist"]
            p.sensing_radius[a] = cfg["predator_radius"] if a < npred else 0.0
            p.capture_radius[a] = 0.0 if a < npred else cfg["capture_radius"]
        p.time_penalty, p.sense_reward, p.capture_reward = cfg["time_penalty"], cfg["sense_reward"], cfg["capture_reward"]
        p.violation_reward = -5.0
    elif scenario == "Warehouse":
        N = int(cfg["n_agents"])
        p.n_agents = N
        p.num_neighbors = int(cfg["num_neighbors"])
        p.obs_dim = 3 * (p.num_neighbors + 1)
        for a in range(N):
            p.agent_step[a] = cfg["step_dist"]
        p.load_reward, p.unload_reward, p.goal_width = cfg["load_reward"], cfg["unload_reward"], cfg["goal_width"]
        p.violation_reward = -5.0
    elif scenario == "Simple":
        N = int(cfg["n_agents"])
        p.n_agents = N
        p.num_prey = 1
        p.obs_dim = 2 * (N + 1)
        for a in range(N):
            p.agent_step[a] = cfg["step_dist"]
        p.reward_scaler = cfg["reward_scaler"]
        p.violation_reward = -5.0
    elif scenario == "ArcticTransport":
        p.n_agents = 4
        p.obs_dim = 30
        p.arctic_normal_step, p.arctic_slow_step, p.arctic_fast_step = \
            cfg["normal_step"], cfg["slow_step"], cfg["fast_step"]
        p.not_reached_penalty, p.dist_multiplier = cfg["not_reached_penalty"], cfg["dist_multiplier"]
        p.violation_reward = -30.0
    else:
        nf, ns = int(cfg["n_fast_agents"]), int(cfg["n_slow_agents"])
        N = int(cfg["n_agents"])
        assert nf + ns == N
        p.n_agents = N
        p.capability_aware = int(bool(cfg["capability_aware"]))
        p.obs_dim = 11 if p.capability_aware else 9
        for a in range(N):
            p.agent_step[a] = cfg["fast_step"] if a < nf else cfg["slow_step"]
            p.torque[a] = int(cfg["small_torque"] if a < nf else cfg["large_torque"])
        p.time_penalty = cfg["time_penalty"]
        p.unload_multiplier, p.load_multiplier = cfg["unload_multiplier"], cfg["load_multiplier"]
        p.end_goal_width, p.zone1_radius = cfg["end_goal_width"], cfg["zone1_radius"]
        p.violation_reward = -6.0
    return p


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


class OracleVecEnv(object):
    """E envs, state in numpy arrays of `dtype` (np.float64 or np.float32)."""

    def __init__(self, scenario, cfg, E, dtype=np.float64, collision_variant="offset"):
        self.scenario, self.cfg, self.E = scenario, dict(cfg), E
        self.dtype = np.dtype(dtype)
        assert self.dtype in (np.dtype(np.float64), np.dtype(np.float32))
        self.p = params_from_config(scenario, cfg, collision_variant, self.dtype)
        N, P, D = self.p.n_agents, self.p.num_prey, self.p.obs_dim
        self.N, self.P, self.D = N, P, D
        f = self.dtype
        self.poses = np.zeros((E, 3, N), f)
        self.carry = np.zeros((E, N), f)
        self.steps = np.zeros(E, np.int32)
        self.prey_loc = np.zeros((E, max(P, 1), 2), f)
        self.prey_sensed = np.zeros((E, max(P, 1)), np.uint8)
        self.prey_captured = np.zeros((E, max(P, 1)), np.uint8)
        self.loaded = np.zeros((E, N), np.uint8)
        self.load = np.zeros((E, N), np.int32)
        self.zone_load = np.zeros((E, 2), np.int32)
        self.messages = np.zeros((E, 4), np.int32)
        self.grid = np.zeros((E, 96), np.uint8)
        self.goal_col = np.zeros(E, np.int32)
        self.pixel_type = np.zeros((E, N), np.uint8)
        self.reached_goal = np.zeros((E, N), np.uint8)
        self.obs = np.zeros((E, N, D), f)
        self.reward = np.zeros((E, N), f)
        self.done = np.zeros(E, np.uint8)
        self.dist = np.zeros((E, N), f)
        self.viol = np.zeros(E, np.uint8)
        self.remaining = np.zeros(E, np.int32)
        self.qp_sweeps = np.zeros(E, np.int32)
        self._fn = lib().orc_step_f64 if f == np.float64 else lib().orc_step_f32
        self._fn.restype = C.c_int

    STATE_KEYS = ("poses", "carry", "steps", "prey_loc", "prey_sensed", "prey_captured", "loaded", "load",
                  "zone_load", "messages", "grid", "goal_col", "pixel_type", "reached_goal")

    def set_state(self, e, **kw):
        for k, v in kw.items():
            arr = getattr(self, k)
            arr[e] = np.asarray(v).astype(arr.dtype)

    def get_state(self, e):
        return {k: getattr(self, k)[e].copy() for k in self.STATE_KEYS}

    def step(self, actions, threads=1):
        """threads > 1: the envs are independent, so contiguous env slices are stepped by a thread pool (the C call
        releases the GIL) -- used by the parity tests at the BASELINE batch sizes."""
        actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.E, self.N)
        if threads > 1 and self.E >= 2 * threads:
            from concurrent.futures import ThreadPoolExecutor
            cuts = [self.E * i // threads for i in range(threads + 1)]
            with ThreadPoolExecutor(threads) as pool:
                list(pool.map(lambda i: self._step_slice(actions, cuts[i], cuts[i + 1]), range(threads)))
        else:
            self._step_slice(actions, 0, self.E)
        return self.obs, self.reward, self.done, {"dist_travelled": self.dist, "violation": self.viol,
                                                  "remaining": self.remaining}

    def _step_slice(self, actions, lo, hi):
        ct = C.c_double if self.dtype == np.float64 else C.c_float

        class St(C.Structure):
            _fields_ = [("poses", C.POINTER(ct)), ("carry", C.POINTER(ct)), ("steps", C.POINTER(C.c_int32)),
                        ("prey_loc", C.POINTER(ct)), ("prey_sensed", C.POINTER(C.c_uint8)),
                        ("prey_captured", C.POINTER(C.c_uint8)), ("loaded", C.POINTER(C.c_uint8)),
                        ("load", C.POINTER(C.c_int32)), ("zone_load", C.POINTER(C.c_int32)),
                        ("messages", C.POINTER(C.c_int32)), ("grid", C.POINTER(C.c_uint8)),
                        ("goal_col", C.POINTER(C.c_int32)), ("pixel_type", C.POINTER(C.c_uint8)),
                        ("reached_goal", C.POINTER(C.c_uint8))]

        class Out(C.Structure):
            _fields_ = [("obs", C.POINTER(ct)), ("reward", C.POINTER(ct)), ("done", C.POINTER(C.c_uint8)),
                        ("dist", C.POINTER(ct)), ("viol", C.POINTER(C.c_uint8)),
                        ("remaining", C.POINTER(C.c_int32)), ("qp_sweeps", C.POINTER(C.c_int32))]

        def sl(a, t):   # the slice's rows of an [E, ...] array (contiguous: the env index is the leading dimension)
            return _ptr(a[lo:hi], t)

        st = St(sl(self.poses, ct), sl(self.carry, ct), sl(self.steps, C.c_int32), sl(self.prey_loc, ct),
                sl(self.prey_sensed, C.c_uint8), sl(self.prey_captured, C.c_uint8), sl(self.loaded, C.c_uint8),
                sl(self.load, C.c_int32), sl(self.zone_load, C.c_int32), sl(self.messages, C.c_int32),
                sl(self.grid, C.c_uint8), sl(self.goal_col, C.c_int32), sl(self.pixel_type, C.c_uint8),
                sl(self.reached_goal, C.c_uint8))
        out = Out(sl(self.obs, ct), sl(self.reward, ct), sl(self.done, C.c_uint8), sl(self.dist, ct),
                  sl(self.viol, C.c_uint8), sl(self.remaining, C.c_int32), sl(self.qp_sweeps, C.c_int32))
        rc = self._fn(C.byref(self.p), C.c_int(hi - lo), C.byref(st), sl(actions, C.c_int32), C.byref(out))
        if rc != 0:
            raise RuntimeError(f"orc_step failed: {rc}")


def spec_sincos_f32(t):
    t = np.ascontiguousarray(t, np.float32)
    s, c = np.empty_like(t), np.empty_like(t)
    lib().orc_sincos_f32(C.c_int(t.size), _ptr(t, C.c_float), _ptr(s, C.c_float), _ptr(c, C.c_float))
    return s, c


def spec_atan2_f32(y, x):
    y = np.ascontiguousarray(y, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    o = np.empty_like(y)
    lib().orc_atan2_f32(C.c_int(y.size), _ptr(y, C.c_float), _ptr(x, C.c_float), _ptr(o, C.c_float))
    return o


def controller(scenario, cfg, poses, goals, dtype=np.float64, collision_variant="offset"):
    """a3-a8 for one env: poses 3xN, goals 2xN -> (dxu 2xN after set_velocities clipping, sweeps)."""
    p = params_from_config(scenario, cfg, collision_variant, dtype)
    f = np.dtype(dtype)
    ct = C.c_double if f == np.float64 else C.c_float
    poses = np.ascontiguousarray(poses, f)
    goals = np.ascontiguousarray(goals[:2], f)
    p.n_agents = poses.shape[1]
    dxu = np.zeros((2, poses.shape[1]), f)
    fn = lib().orc_controller_f64 if f == np.float64 else lib().orc_controller_f32
    fn.restype = C.c_int
    sweeps = fn(C.byref(p), _ptr(poses, ct), _ptr(goals, ct), _ptr(dxu, ct))
    return dxu, sweeps


class OrcGrid(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("spacing", C.c_float), ("w2", C.c_float),
                ("h2", C.c_float), ("ox1", C.c_float), ("ox2", C.c_float), ("oy1", C.c_float),
                ("oy2", C.c_float)]


class OrcResetParams(C.Structure):
    _fields_ = [("scenario", C.c_int32), ("n_agents", C.c_int32), ("num_prey", C.c_int32),
                ("keep_theta", C.c_int32), ("agent_grid", OrcGrid), ("prey_grid", OrcGrid),
                ("zone1_mean", C.c_float), ("zone1_std", C.c_float), ("zone2_mean", C.c_float),
                ("zone2_std", C.c_float)]


def reset_env_f32(rp, seed, global_env, episode):
    """The float-spec reset sampler for one env: returns (poses [3,N], prey_loc [P,2], zone_load [2])."""
    N, P = rp.n_agents, max(rp.num_prey, 1)
    poses = np.zeros((3, N), np.float32)
    prey = np.zeros((P, 2), np.float32)
    zone = np.zeros(2, np.int32)
    fn = lib().orc_reset_env_f32
    fn.restype = None
    fn(C.byref(rp), C.c_uint64(seed), C.c_uint64(global_env), C.c_uint32(episode), _ptr(poses, C.c_float),
       _ptr(prey, C.c_float), _ptr(zone, C.c_int32))
    return poses, prey, zone


def reset_arctic_f32(seed, global_env, episode):
    """ArcticTransport reset twin: (poses [3,4], grid [96] u8, goal_col)."""
    poses = np.zeros((3, 4), np.float32)
    grid = np.zeros(96, np.uint8)
    gc = C.c_int32(0)
    fn = lib().orc_reset_arctic_f32
    fn.restype = None
    fn(C.c_uint64(seed), C.c_uint64(global_env), C.c_uint32(episode), _ptr(poses, C.c_float),
       _ptr(grid, C.c_uint8), C.byref(gc))
    return poses, grid, gc.value
