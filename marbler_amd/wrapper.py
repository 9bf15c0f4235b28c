"""Single-env facade with the reference's exact Python-level surface (robotarium_gym/wrapper.py).

`Wrapper(env_name, config_path)` is what the gym ids `robotarium_gym:<Scenario>-v0` construct
(__init__.py:4-23).  It returns what the reference returns, type for type:
    reset() -> list of N lists of zeros                       (PredatorCapturePrey.py:136)
    step(action_n) -> (tuple of N float64 arrays [D], list of N rewards, list of N bools,
                       info dict: 'dist_travelled' float64[N], 'message' str if a violation
                       ended the episode, 'remaining' int when the scenario reports it)
The arithmetic runs on the GPU through VecRobotariumEnv with E = 1; this class exists for
drop-in compatibility (EPyMARL's gymma wrapper, evaluation scripts), not for throughput.
"""
import numpy as np
import torch

from .params import default_config_path
from .spaces import scenario_spaces
from .vec_env import VIOLATION_MESSAGES, VecRobotariumEnv

SCENARIOS = ("PredatorCapturePrey", "Warehouse", "MaterialTransport", "Simple", "ArcticTransport")


class _ScenarioFacade(object):
    """What `Wrapper.env` exposes in the reference (scenarios/base.py): num_robots, agent_poses,
    args, get_action_space / get_observation_space, reset, step."""

    def __init__(self, env_name, config_path, device):
        self.vec = VecRobotariumEnv(env_name, 1, config_path=config_path, device=device, auto_reset=False,
                                    reference_reset_obs=True)
        cfg = self.vec.cfg
        if cfg.get("seed", -1) != -1:
            self.vec.seed = int(cfg["seed"])
        self.args = type("objectview", (), dict(cfg))()
        self.num_robots = self.vec.N
        self.action_space, self.observation_space = scenario_spaces(env_name, self.vec.params)
        self._scenario = env_name
        # host side of a step: actions go up through one pinned buffer, everything a step returns comes
        # down with ONE copy of the env's output allocation (vec_env.py: the outputs are views of it)
        v = self.vec
        self._act_host = torch.zeros(1, v.N, dtype=torch.int32).pin_memory()
        self._act_dev = torch.zeros(1, v.N, dtype=torch.int32, device=v.device)
        self._out_host = torch.zeros_like(v._out_arena, device="cpu").pin_memory()
        o, n, N, D = v._out_offsets, self._out_host.numpy(), v.N, v.D
        self._h_obs = n[o["obs"]:o["obs"] + N * D * 4].view(np.float32).reshape(N, D)
        self._h_rew = n[o["reward"]:o["reward"] + N * 4].view(np.float32)
        self._h_dist = n[o["dist_travelled"]:o["dist_travelled"] + N * 4].view(np.float32)
        self._h_rem = n[o["remaining"]:o["remaining"] + 4].view(np.int32)
        self._h_done = n[o["done_u8"]:o["done_u8"] + 1]
        self._h_viol = n[o["violation"]:o["violation"] + 1]

    @property
    def agent_poses(self):
        return self.vec.poses[0].double().cpu().numpy()          # 3 x N, like the reference

    def get_action_space(self):
        return self.action_space

    def get_observation_space(self):
        return self.observation_space

    def reset(self):
        self.vec.reset()
        return [[0] * self.vec.D] * self.num_robots

    def step(self, actions_):
        v = self.vec
        self._act_host.numpy()[0, :] = np.asarray(actions_, dtype=np.int32).reshape(self.num_robots)
        self._act_dev.copy_(self._act_host, non_blocking=True)
        rc = v.step_raw(self._act_dev.data_ptr())
        if rc != 0:
            from . import _lib
            _lib.check(rc, "rg_step")
        self._out_host.copy_(v._out_arena, non_blocking=True)
        torch.cuda.current_stream(v.device).synchronize()
        obs = self._h_obs.astype(np.float64)
        terminated = bool(self._h_done[0])
        out = {}
        viol = int(self._h_viol[0])
        if viol:
            out["message"] = VIOLATION_MESSAGES[viol]
        rem = int(self._h_rem[0])
        if rem >= 0:
            out["remaining"] = rem
        out["dist_travelled"] = self._h_dist.astype(np.float64)
        return [obs[i] for i in range(self.num_robots)], [float(r) for r in self._h_rew], \
            [terminated] * self.num_robots, out

    def render(self, mode='human'):
        pass


env_dict = {name: name for name in SCENARIOS}  # wrapper.py:12-16


class Wrapper(object):
    def __init__(self, env_name, config_path=None, device="cuda:0"):
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
    for name in SCENARIOS:
        register(name + "-v0", entry_point=entry_point,
                 kwargs={"env_name": name, "config_path": default_config_path(name)})
        ids.append(name + "-v0")
    return ids
