"""Single-env facade with the reference's exact Python-level surface (robotarium_gym/wrapper.py).

`Wrapper(env_name, config_path)` is what the gym ids `robotarium_gym:<Scenario>-v0` construct
(__init__.py:4-23).  It returns what the reference returns, type for type:
    reset() -> list of N lists of zeros                       (PredatorCapturePrey.py:136)
    step(action_n) -> (tuple of N float64 arrays [D], list of N rewards, list of N bools,
                       info dict: 'dist_travelled' float64[N], 'message' str if a violation
                       ended the episode, 'remaining' int when the scenario reports it)
The arithmetic runs on the GPU through VecRobotariumEnv with E = 1; this class exists for
drop-in compatibility (EPyMARL's gymma wrapper, evaluation scripts), not for throughput.

Seeding follows the reference (PredatorCapturePrey.py:27-28 and the other constructors): `seed: s` in the
config seeds ONE legacy NumPy stream at construction from which every reset() draws, with the reference's
own call sequence (marbler_amd/reference_reset.py), so the episodes start exactly where the reference's
`Wrapper` starts them; `seed: -1` (the default of every shipped YAML) means "do not seed": the device
sampler gets a fresh key from os.urandom, so no two instances -- e.g. the `batch_size_run` copies
EPyMARL's parallel runner builds from one key -- replay the same initial conditions.

`Wrapper` derives from gym.Env (or gymnasium.Env) when one is installed, like the reference's
(wrapper.py:19): `gym.make('robotarium_gym:<Scenario>-v0')` sets `env.unwrapped.spec`, and gym's
TimeLimit / other wrappers read `metadata`, `reward_range`, `spec`.
"""
import random as _pyrandom
import time as _time

import numpy as np
import torch

try:  # the reference's base class (wrapper.py:1,19)
    from gym import Env as _EnvBase
except Exception:  # noqa: BLE001
    try:
        from gymnasium import Env as _EnvBase
    except Exception:  # noqa: BLE001
        _EnvBase = object

from .params import default_config_path
from .spaces import scenario_spaces
from .vec_env import VIOLATION_MESSAGES, VecRobotariumEnv

SCENARIOS = ("PredatorCapturePrey", "Warehouse", "MaterialTransport", "Simple", "ArcticTransport")


class _ScenarioFacade(object):
    """What `Wrapper.env` exposes in the reference (scenarios/base.py): num_robots, agent_poses,
    args, get_action_space / get_observation_space, reset, step."""

    def __init__(self, env_name, config_path, device):
        # `real_time: True` (rps sim_in_real_time: every simulator iteration waits until 0.033 s have passed since the one before,
        # rps Robotarium.step) is a property of this single-env facade, not of the batched engine: the step is computed at once
        # and step() returns when the reference's would have -- update_frequency x 0.033 s after it was called.  (A step that
        # ends early in a violation is paced like a full one: the exit iteration is not part of what a step returns.)
        from .params import load_config
        self._real_time = bool(load_config(env_name, config_path).get("real_time", False))
        self.vec = VecRobotariumEnv(env_name, 1, config_path=config_path, device=device, auto_reset=False,
                                    reference_reset_obs=True, seed=None,   # seed -1: a fresh key per instance
                                    overrides={"real_time": False} if self._real_time else None)
        cfg = dict(self.vec.cfg, real_time=self._real_time)
        self._rng = self._pyrandom = None
        if cfg.get("seed", -1) != -1:
            self.vec.seed = int(cfg["seed"])
            self._rng = np.random.RandomState(int(cfg["seed"]))   # = np.random.seed(args.seed) in the scenario ctor
            # ArcticTransport draws its goal column from Python's `random`, which the reference never seeds:
            # that one draw is made reproducible here (derived from the config seed)
            self._pyrandom = _pyrandom.Random(int(cfg["seed"]))
        self.args = type("objectview", (), dict(cfg))()
        self.num_robots = self.vec.N
        self.action_space, self.observation_space = scenario_spaces(env_name, self.vec.params)
        self._scenario = env_name

    @property
    def agent_poses(self):
        return self.vec.poses[0].double().cpu().numpy()          # 3 x N, like the reference

    def get_action_space(self):
        return self.action_space

    def get_observation_space(self):
        return self.observation_space

    def reset(self):
        self.vec.reset(reference_rng=self._rng, py_random=self._pyrandom)
        return [[0] * self.vec.D] * self.num_robots

    def step(self, actions_):
        t0 = _time.monotonic()
        h_obs, h_rew, terminated, viol, rem, h_dist = self.vec.host_step(actions_)
        if self._real_time:
            wait = self.vec.params.update_frequency * self.vec.params.time_step - (_time.monotonic() - t0)
            if wait > 0:
                _time.sleep(wait)
        obs = h_obs.astype(np.float64)
        out = {}
        if viol and self._scenario == "Simple":
            out["remaining"] = VIOLATION_MESSAGES[viol]          # simple.py:176 files the message under 'remaining'
        elif viol:
            out["message"] = VIOLATION_MESSAGES[viol]
        if rem >= 0:
            out["remaining"] = rem
        out["dist_travelled"] = h_dist.astype(np.float64)
        return [obs[i] for i in range(self.num_robots)], [float(r) for r in h_rew], \
            [terminated] * self.num_robots, out

    def render(self, mode='human'):
        pass


env_dict = {name: name for name in SCENARIOS}  # wrapper.py:12-16


class Wrapper(_EnvBase):
    metadata = {"render.modes": [], "render_modes": []}
    reward_range = (-float("inf"), float("inf"))
    spec = None

    def __init__(self, env_name, config_path=None, device="cuda:0"):
        if _EnvBase is not object:
            super().__init__()
        if env_name not in env_dict:
            raise KeyError(f"scenario {env_name!r} is not built (have {sorted(env_dict)})")
        self.env = _ScenarioFacade(env_name, config_path or default_config_path(env_name), device)
        self.observation_space = self.get_observation_space()
        self.action_space = self.get_action_space()
        self.n_agents = self.env.num_robots

    def reset(self):
        return self.env.reset()

    def step(self, action_n):
        obs_n, reward_n, done_n, info_n = self.env.step(action_n)
        return tuple(obs_n), reward_n, done_n, info_n

    def get_action_space(self):
        return self.env.get_action_space()

    def get_observation_space(self):
        return self.env.get_observation_space()

    def render(self, mode='human'):
        pass

    def close(self):
        self.env.vec.close()


def register_gym_ids(entry_point="marbler_amd.wrapper:Wrapper"):
    """Registers `<Scenario>-v0` like robotarium_gym/__init__.py:4-23 when gym/gymnasium exists.
    Returns the list of ids registered (empty if no gym is installed)."""
    try:
        from gym.envs.registration import register
    except Exception:  # noqa: BLE001
        try:
            from gymnasium.envs.registration import register
        except Exception:  # noqa: BLE001
            return []
    ids = []
    for name in ("PredatorCapturePrey", "Warehouse", "Simple", "ArcticTransport", "MaterialTransport"):   # __init__.py:4-10
        register(name + "-v0", entry_point=entry_point,
                 kwargs={"env_name": name, "config_path": default_config_path(name)})
        ids.append(name + "-v0")
    return ids
