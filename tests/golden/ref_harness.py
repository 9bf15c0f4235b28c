"""Drive the reference's OWN Python (Wrapper -> scenario -> roboEnv -> Controller) in this
container, with the restated `rps` (oracle/rps_restated) injected for the absent third-party
simulator and inert stand-ins for gym / tensorflow / imageio.

Runs ONLY where /root/reference exists (the build container).  Nothing here is imported by
the product, by `-m gpu` tests, smoke() or bench.py; the vectors it produces are committed
as tests/golden/*.npz by make_golden.py.
"""
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import yaml

REFERENCE_ROOT = os.environ.get("MARBLER_REFERENCE", "/root/reference")
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "robotarium_gym"))


def _install_stubs():
    sys.dont_write_bytecode = True
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")
        spaces = types.ModuleType("gym.spaces")

        class _Space(object):
            def __init__(self, *a, **k):
                self.args = a
                self.kwargs = k

        class Discrete(_Space):
            @property
            def n(self):
                return self.args[0]

        class Box(_Space):
            @property
            def shape(self):
                return self.kwargs.get("shape")

        class Tuple(_Space):
            @property
            def spaces(self):
                return self.args[0]

            def __len__(self):
                return len(self.args[0])

            def __getitem__(self, i):
                return self.args[0][i]

        class Env(object):
            def __init__(self):
                pass

        spaces.Discrete, spaces.Box, spaces.Tuple = Discrete, Box, Tuple
        gym.spaces = spaces
        gym.Env = Env
        envs = types.ModuleType("gym.envs")
        reg = types.ModuleType("gym.envs.registration")
        reg.register = lambda *a, **k: None
        envs.registration = reg
        gym.envs = envs
        sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.envs": envs,
                            "gym.envs.registration": reg})
    for name in ("tensorflow", "imageio"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    rps_dir = os.path.join(REPO_ROOT, "oracle", "rps_restated")
    if rps_dir not in sys.path:
        sys.path.insert(0, rps_dir)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


_SCENARIO_DIR = {"PredatorCapturePrey": "PredatorCapturePrey", "Warehouse": "Warehouse",
                 "MaterialTransport": "MaterialTransport", "Simple": "Simple", "ArcticTransport": "ArcticTransport"}


def load_reference_config(scenario):
    path = os.path.join(REFERENCE_ROOT, "robotarium_gym", "scenarios", _SCENARIO_DIR[scenario], "config.yaml")
    with open(path) as f:
        return yaml.safe_load(f)


def make_reference_wrapper(scenario, overrides, collision_variant="offset"):
    """Returns (wrapper, config_dict).  Stdout of the reference is swallowed (it prints on
    termination)."""
    _install_stubs()
    import random as _pyrandom
    _pyrandom.seed(int(overrides.get("seed", 0)) + 12345)   # ArcticTransport.py:72 draws from Python's `random`
    import rps.robotarium as rr
    import rps.utilities.barrier_certificates as bc
    rr.COLLISION_VARIANT = collision_variant
    rr._ERRORS.clear()
    # which solver stands in for cvxopt below rps' certificate closures (oracle/rps_restated): the exact projection (sim_spec_v0's
    # default) or the restated interior-point `qp` at rps' options -- what the reference's stack itself evaluates
    solver = overrides.get("barrier_solver", "exact")
    assert solver in ("exact", "cvxopt"), solver
    bc.QP_SOLVER = solver
    cfg = load_reference_config(scenario)
    cfg.update({"show_figure_frequency": -1, "enable_logging": False, "save_gif": False,
                "real_time": False, "robotarium": False})
    cfg.update(overrides)
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(cfg, f)
        path = f.name
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            from robotarium_gym.wrapper import Wrapper
            w = Wrapper(scenario, path)
    finally:
        os.unlink(path)
    family = {k: cfg[k] for k in ("safety_radius", "barrier_gain", "unsafe_barrier_gain", "magnitude_limit") if k in cfg}
    if family:
        # a certificate with other arguments: the reference's own route is Controller(type='custom', custom=<closure>)
        # (utilities/controller.py:17-18), the closure being one of rps' factories called with those arguments
        from robotarium_gym.utilities.controller import Controller
        kw = {"barrier_gain": family.get("barrier_gain", 100), "magnitude_limit": family.get("magnitude_limit", 0.2)}
        if cfg.get("barrier_certificate", "safe") == "safe":
            cert = bc.create_single_integrator_barrier_certificate2(unsafe_barrier_gain=family.get("unsafe_barrier_gain", 1e6),
                                                                    safety_radius=family.get("safety_radius", 0.2), **kw)
        else:
            cert = bc.create_single_integrator_barrier_certificate(safety_radius=family.get("safety_radius", 0.17), **kw)
        w.env.env.controller = Controller("custom", custom=cert)
    return w, cfg


_MSG = {"": 0, "collision": 1, "boundary": 2, "collision_boundary": 3}


def quiet_reset(w):
    with contextlib.redirect_stdout(io.StringIO()):
        return w.reset()


def quiet_step(w, actions):
    if hasattr(w.env, "prey_locs"):
        # NumPy-2 incompatibility at PredatorCapturePrey.py:185 (`ndarray == []`); prey_locs is
        # dead state, so it is cleared from outside and the reference stays untouched.
        w.env.prey_locs = []
    with contextlib.redirect_stdout(io.StringIO()):
        return w.step(actions)


def snapshot_state(w, scenario):
    """Full env state as the build represents it (see DESIGN.md 'state')."""
    s = w.env
    ro = s.env  # roboEnv
    poses = np.array(s.agent_poses, dtype=np.float64)
    if ro.previous_pose is None:
        carry = np.zeros(poses.shape[1])
    else:
        carry = np.linalg.norm(poses[:2, :] - ro.previous_pose[:2, :], axis=0)
    st = {"poses": poses.copy(), "carry": carry, "steps": np.int64(s.episode_steps)}
    if scenario == "PredatorCapturePrey":
        st["prey_loc"] = np.array(s.prey_loc, dtype=np.float64).copy()
        st["prey_sensed"] = np.array(s.prey_sensed, dtype=np.uint8)
        st["prey_captured"] = np.array(s.prey_captured, dtype=np.uint8)
    elif scenario == "Warehouse":
        st["loaded"] = np.array([a.loaded for a in s.agents], dtype=np.uint8)
    elif scenario == "MaterialTransport":
        st["load"] = np.array([a.load for a in s.agents], dtype=np.int64)
        st["zone_load"] = np.array([s.zone1_load, s.zone2_load], dtype=np.int64)
        st["messages"] = np.array(s.messages, dtype=np.int64)
    elif scenario == "Simple":
        st["prey_loc"] = np.array(s.goal_loc, dtype=np.float64).reshape(1, 2).copy()   # the goal, as a 1-prey block
    elif scenario == "ArcticTransport":
        st["grid"] = np.array(s.grid, dtype=np.uint8).reshape(-1).copy()
        st["goal_col"] = np.int64(s.goal_loc[1])
        st["pixel_type"] = np.array([a.pixel_type for a in s.agents], dtype=np.uint8)
        st["reached_goal"] = np.array([a.reached_goal for a in s.agents], dtype=np.uint8)
    return st


def step_record(w, scenario, actions):
    """One reference step; returns dict of pre-state, inputs, outputs, post-state."""
    pre = snapshot_state(w, scenario)
    obs, rew, done, info = quiet_step(w, list(int(a) for a in actions))
    post = snapshot_state(w, scenario)
    rec = {"pre_" + k: v for k, v in pre.items()}
    rec.update({"post_" + k: v for k, v in post.items()})
    rec["actions"] = np.array(actions, dtype=np.int64)
    rec["obs"] = np.array([np.asarray(o, dtype=np.float64) for o in obs])
    rec["reward"] = np.array(rew, dtype=np.float64)
    rec["done"] = np.uint8(bool(done[0]))
    assert all(bool(d) == bool(done[0]) for d in done)
    rec["dist"] = np.array(info["dist_travelled"], dtype=np.float64)
    rec["viol"] = np.uint8(_MSG[info.get("message", "")])
    rem = info.get("remaining", -1)
    # Simple puts the violation STRING under 'remaining' (simple.py:176): recorded as the message code
    if isinstance(rem, str):
        rec["viol"] = np.uint8(_MSG[rem])
        rem = -1
    rec["remaining"] = np.int64(rem)
    return rec
