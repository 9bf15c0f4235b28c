"""EPyMARL `gymma`-shaped adapters over VecRobotariumEnv.

The reference trains through EPyMARL's `gymma` env wrapper (README.md:25-28:
`--env-config=gymma with env_args.key="robotarium_gym:<Scenario>-v0" env_args.time_limit=...`),
which is NOT part of /root/reference.  Its consumer contract, as recalled from upstream EPyMARL
(`src/envs/__init__.py::_GymmaWrapper`) and consistent with the README's notes:

  * `gym.make(key)` wrapped in `TimeLimit(max_episode_steps=time_limit)` and `FlattenObservation`;
  * `step(actions) -> (float(sum(reward_n)), all(done_n), info)`;
  * `get_obs()` = per-agent observations padded to the longest; `get_state()` = their concatenation;
  * `get_avail_actions()` = all ones (every action always available), `get_total_actions()` = the
    largest Discrete.n; `get_env_info()` = {state_shape, obs_shape, n_actions, n_agents, episode_limit};
  * `reset()` returns `(get_obs(), get_state())`.

`GymmaVecEnv` is the batched form: one object, E envs, torch tensors, one HIP launch per step --
what EPyMARL's `parallel` runner gets from `batch_size_run` worker processes
(scenarios/*/models/*.json: "runner": "parallel", "batch_size_run": 4-10).  `GymmaEnv` is the
single-env form with EPyMARL's Python types, for dropping into an unmodified EPyMARL tree
(register it under REGISTRY["gymma"]).
"""
import numpy as np
import torch

from .evaluate import explore_select
from .vec_env import VecRobotariumEnv

N_ACTIONS = {"PredatorCapturePrey": 5, "Warehouse": 5, "MaterialTransport": 20, "Simple": 5, "ArcticTransport": 5}


def scenario_from_key(key):
    """'robotarium_gym:PredatorCapturePrey-v0' -> 'PredatorCapturePrey'."""
    name = key.split(":")[-1]
    if not name.endswith("-v0"):
        raise ValueError(f"unknown gym key {key!r}")
    return name[:-3]


class GymmaVecEnv(object):
    """E envs behind gymma's method names; everything is a device tensor with a leading E axis."""

    def __init__(self, key, num_envs, time_limit, config_path=None, overrides=None, device="cuda:0", seed=0,
                 env_offset=0, fused=True, alias_outputs=False):
        """fused: gym's TimeLimit and the gymma reductions run inside the env's step launch (VecRobotariumEnv.
        enable_time_limit): step() is ONE launch.  fused=False composes them from torch ops around the step (a dozen
        launches; kept as the readable statement of the contract and as the check of the fused path).
        alias_outputs: step() returns the env's persistent output buffers themselves (reward sum, terminated,
        TimeLimit.truncated) instead of copies -- one small copy launch less per step, but the NEXT step overwrites
        what the previous one returned; only for callers that consume a step's outputs before stepping again."""
        self.fused = bool(fused)
        self.alias_outputs = bool(alias_outputs)
        self.scenario = scenario_from_key(key)
        self.env = VecRobotariumEnv(self.scenario, num_envs, config_path=config_path, overrides=overrides,
                                    device=device, seed=seed, env_offset=env_offset, auto_reset=True,
                                    reference_reset_obs=True)
        self.E, self.n_agents = self.env.E, self.env.N
        self.episode_limit = int(time_limit)
        self.n_actions = N_ACTIONS[self.scenario]
        self.obs_size = self.env.D
        self._elapsed = torch.zeros(self.E, dtype=torch.int32, device=self.env.device)
        self._avail = torch.ones(self.E, self.n_agents, self.n_actions, dtype=torch.int32, device=self.env.device)
        self._obs = self.env.obs
        # gym's TimeLimit can only fire if it is not longer than the scenario's own horizon
        # (episodes end at step max_episode_steps + 1 at the latest): otherwise nothing to do per step
        self._can_truncate = self.episode_limit <= int(self.env.params.max_episode_steps) + 1
        self._ended = None
        if self.fused:
            self.env.enable_time_limit(self.episode_limit)

    # -- gymma surface, batched
    def reset(self):
        self._obs = self.env.reset()
        self._elapsed.zero_()
        self._ended = None
        return self.get_obs(), self.get_state()

    def step(self, actions):
        """actions [E, N] int -> (reward [E] f32 = sum over agents, terminated [E] bool, info).
        Envs that terminate (scenario rule or time limit) start a new episode; their next observation
        is the reset observation (zeros, as the reference returns from reset()).
        reward, terminated and info["TimeLimit.truncated"] are fresh tensors (a runner may keep step t's while it
        takes step t + 1) unless the env was built with alias_outputs=True; the other info entries (dist_travelled,
        violation, remaining) are always views of the env's output buffers, valid until the next step."""
        obs, reward, done, info = self.env.step(actions)
        if self.fused:      # everything below happened inside that one launch
            self._obs, self._ended = None, self.env.ended
            out = dict(info)
            if self.alias_outputs:
                out["TimeLimit.truncated"] = self.env.truncated
                return self.env.reward_sum, self.env.ended, out
            reward_sum, ended, out["TimeLimit.truncated"] = self.env.gymma_outputs_copy()   # one copy launch for the three
            return reward_sum, ended, out
        self._elapsed += 1
        truncated = (self._elapsed >= self.episode_limit) & ~done          # gym TimeLimit
        ended = done | truncated
        # the kernel has already reset `done` envs; gymma users see zeros after a reset (the reference's reset obs)
        self._obs = torch.where(ended[:, None, None], torch.zeros_like(obs), obs)
        self._terminal_obs = obs.clone() if self._can_truncate else obs
        if self._can_truncate:
            # masked launch, no host round trip (waves without a flagged env exit at once); the truncated episodes
            # are booked into the episode statistics like the ones the scenario ended
            self.env.reset(truncated, book_episode=True)
        self._elapsed.masked_fill_(ended, 0)
        out = dict(info)
        out["TimeLimit.truncated"] = truncated
        return reward.sum(dim=1), ended, out

    def get_obs(self):
        if self._obs is None:   # fused step: the next observation of an env that just ended is its reset observation (zeros)
            self._obs = torch.where(self._ended[:, None, None], 0.0, self.env.obs)
        return self._obs

    def get_obs_agent(self, agent_id):
        return self.get_obs()[:, agent_id]

    def get_obs_size(self):
        return self.obs_size

    def get_state(self):
        return self.get_obs().reshape(self.E, -1)

    def get_state_size(self):
        return self.n_agents * self.obs_size

    def get_avail_actions(self):
        return self._avail

    def get_avail_agent_actions(self, agent_id):
        return self._avail[:, agent_id]

    def get_total_actions(self):
        return self.n_actions

    def get_env_info(self):
        return {"state_shape": self.get_state_size(), "obs_shape": self.get_obs_size(),
                "n_actions": self.get_total_actions(), "n_agents": self.n_agents,
                "episode_limit": self.episode_limit}

    def get_stats(self):
        s, n, t = self.env.episode_stats()
        return {"return_sum": float(s), "episodes": int(n), "steps": int(t)}

    def render(self):
        pass

    def close(self):
        self.env.close()

    def seed(self, seed=None):
        if seed is not None:
            self.env.seed = int(seed)
        return self.env.seed

    def save_replay(self):
        pass


class GymmaEnv(object):
    """One env with EPyMARL's Python-level types (lists of numpy arrays, floats, bools)."""

    def __init__(self, key, time_limit, pretrained_wrapper=None, seed=None, device="cuda:0", **kwargs):
        self._v = GymmaVecEnv(key, 1, time_limit, device=device, seed=seed, overrides=kwargs or None, fused=False)
        # auto-reset is the runner's job in EPyMARL: keep the terminal state until reset() is called
        self._v.env.auto_reset = False
        self.n_agents = self._v.n_agents
        self.episode_limit = self._v.episode_limit
        self._obs_host = np.zeros((self.n_agents, self._v.obs_size), np.float32)
        self._elapsed_host = 0

    def step(self, actions):
        obs, reward, done, viol, _, _ = self._v.env.host_step([int(x) for x in actions])   # one upload, one download
        self._obs_host = obs.copy()
        self._elapsed_host += 1
        truncated = self._elapsed_host >= self.episode_limit and not done
        out = {}
        if truncated:
            out["TimeLimit.truncated"] = True
        if viol:
            from .vec_env import VIOLATION_MESSAGES
            # simple.py:176 files the violation string under 'remaining'
            out["remaining" if self._v.scenario == "Simple" else "message"] = VIOLATION_MESSAGES[viol]
        return float(reward.sum()), done or truncated, out

    def reset(self):
        self._v.env.reset()
        self._obs_host = np.zeros((self.n_agents, self._v.obs_size), np.float32)   # the reference's reset() observation
        self._elapsed_host = 0
        return self.get_obs(), self.get_state()

    def get_obs(self):
        return [self._obs_host[i] for i in range(self.n_agents)]

    def get_obs_agent(self, agent_id):
        return self._obs_host[agent_id]

    def get_obs_size(self):
        return self._v.get_obs_size()

    def get_state(self):
        return self._obs_host.reshape(-1)

    def get_state_size(self):
        return self._v.get_state_size()

    def get_avail_actions(self):
        return [[1] * self._v.n_actions for _ in range(self.n_agents)]

    def get_avail_agent_actions(self, agent_id):
        return [1] * self._v.n_actions

    def get_total_actions(self):
        return self._v.n_actions

    def get_env_info(self):
        return self._v.get_env_info()

    def render(self):
        pass

    def close(self):
        self._v.close()

    def seed(self, seed=None):
        return self._v.seed(seed)

    def save_replay(self):
        pass

    def get_stats(self):
        return {}


class BatchedRunner(object):
    """The data-collection half of EPyMARL's `parallel` runner (`runners/parallel_runner.py::run`, external to
    the reference; its consumer contract is the gymma one above) for E envs at once: a recurrent actor picks
    epsilon-greedy actions from the padded observations (+ agent id), the envs step, and the transition
    tensors EPyMARL stores per time step -- obs, state, avail_actions, actions, reward, terminated -- come
    back time-major with a leading [T] axis, all on the device.  Envs that finish restart inside the
    rollout (the hidden state of their agents restarts at zero), so there is no padding to an episode
    limit: `terminated[t, e]` marks episode ends, `episode_start[t, e]` the first step of an episode."""

    def __init__(self, venv, actor, epsilon=0.0, obs_agent_id=True, seed=0):
        self.venv, self.actor = venv, actor
        self.epsilon = float(epsilon)
        self.obs_agent_id = bool(obs_agent_id)
        dev = venv.env.device
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed))
        self.hidden = actor.init_hidden(venv.E).to(dev)
        self._restart = torch.ones(venv.E, dtype=torch.uint8, device=dev)
        self._q = torch.empty(venv.E, venv.n_agents, venv.n_actions, device=dev)
        self._zero = torch.zeros((), device=dev)
        venv.reset()

    @torch.no_grad()
    def run(self, T):
        """With the fused env step and the fused actor a time step is TWO launches, with exploration too: the actor reads the
        previous step's episode-end flags and observation straight from the batch and writes its (epsilon-)greedy actions into it; the env step reads
        those actions and writes the next observation (zeros for an env that ended: the reset observation), the summed reward
        and the episode-end flags into the batch (VecRobotariumEnv.step_into).  `state` is a view of `obs` ([E, N * D] of the
        same memory: gymma's state IS the concatenated observations); `episode_start` is filled once per call."""
        v, env, dev = self.venv, self.venv.env, self.venv.env.device
        E, N, D, A = v.E, v.n_agents, v.obs_size, v.n_actions
        if int(T) < 1:
            raise ValueError("run(T) collects T >= 1 time steps")
        out = {"obs": torch.empty(T + 1, E, N, D, device=dev),
               "avail_actions": torch.ones(T + 1, E, N, A, dtype=torch.int32, device=dev),
               "actions": torch.empty(T, E, N, dtype=torch.int32, device=dev),
               "reward": torch.empty(T, E, device=dev), "terminated": torch.empty(T, E, dtype=torch.bool, device=dev),
               "episode_start": torch.empty(T, E, dtype=torch.bool, device=dev)}
        fused = self.actor.fused_supported()
        eye = torch.eye(N, device=dev).unsqueeze(0).expand(E, N, N)
        direct = fused and v.fused
        eps = self.epsilon
        # exploration: ONE uniform per agent and time step, drawn for the whole call in one launch (evaluate.explore_select is the rule)
        u_all = torch.rand(T, E, N, generator=self.gen, device=dev) if eps > 0.0 else None
        if direct:
            out["state"] = out["obs"].view(T + 1, E, N * D)
            out["obs"][0] = v.get_obs()            # zeros right after a reset, like the reference's reset()
            out["episode_start"][0] = self._restart.view(torch.bool)
            term_u8 = out["terminated"].view(torch.uint8)
            env._sync_stream()
            for t in range(T):
                obs = out["obs"][t]
                restart = self._restart if t == 0 else term_u8[t - 1]
                self.actor.forward_fused(obs, self.hidden, append_agent_id=self.obs_agent_id, restart=restart, q_out=self._q,
                                         actions_out=out["actions"][t], explore_u=None if u_all is None else u_all[t],
                                         epsilon=eps)
                rc = env.step_into(out["actions"][t].data_ptr(), out["obs"][t + 1].data_ptr(), out["reward"][t].data_ptr(),
                                   term_u8[t].data_ptr())
                if rc != 0:
                    from . import _lib
                    _lib.check(rc, "rg_step")
            if T > 1:
                out["episode_start"][1:] = out["terminated"][:-1]
            # the wrapper's own view of "the current observation" and of which envs just restarted, for whoever steps next
            self._restart = term_u8[T - 1].clone()
            v._obs, v._ended = out["obs"][T].clone(), out["terminated"][T - 1].clone()
            return out
        out["state"] = torch.empty(T + 1, E, N * D, device=dev)
        for t in range(T):
            obs = v.get_obs()                  # zeros right after a reset, like the reference's reset()
            out["obs"][t] = obs
            out["state"][t] = obs.reshape(E, N * D)
            out["episode_start"][t] = self._restart.view(torch.bool)
            if fused:
                greedy = out["actions"][t] if self.epsilon <= 0.0 else None
                _, greedy = self.actor.forward_fused(obs.contiguous(), self.hidden, append_agent_id=self.obs_agent_id,
                                                     restart=self._restart, q_out=self._q, actions_out=greedy)
            else:
                h_in = torch.where(self._restart.bool()[:, None, None], torch.zeros_like(self.hidden), self.hidden)
                q, h = self.actor.forward(torch.cat([obs, eye], dim=2) if self.obs_agent_id else obs, h_in)
                self.hidden.copy_(h)
                greedy = q.argmax(dim=2).to(torch.int32)
            if eps > 0.0:
                explore_select(greedy, u_all[t], eps, A, out=out["actions"][t])
            elif not fused:
                out["actions"][t] = greedy
            reward, ended, _ = v.step(out["actions"][t])
            out["reward"][t] = reward
            out["terminated"][t] = ended
            self._restart.copy_(ended)
        out["obs"][T] = v.get_obs()
        out["state"][T] = v.get_state()
        return out
