"""VecRobotariumEnv -- E independent Robotarium-gym envs stepped by one HIP launch.

Host-side mirror of the reference's env surface (wrapper.py:19-50 over scenarios/<S>):
`reset()`, `step(actions)`, plus `get_obs()` (the name EPyMARL's gymma layer uses), all on
torch-ROCm tensors.  State lives in HBM as torch tensors this object owns; the kernels
(librobogym_hip.so, include/robogym.h) read and write them in place.

Shapes (E envs, N agents, D per-agent obs length):
    obs [E,N,D] f32 | reward [E,N] f32 | done [E] bool
    info: dist_travelled [E,N] f32, violation [E] u8 (0 '', 1 collision, 2 boundary,
          3 collision_boundary = info['message']), remaining [E] i32 (-1 = key absent)
"""
import ctypes as C

import torch

from . import _lib
from .params import load_config, make_params

VIOLATION_MESSAGES = ("", "collision", "boundary", "collision_boundary")  # roboEnv.py:82-94


class VecRobotariumEnv(object):
    def __init__(self, scenario, num_envs, config_path=None, overrides=None, device="cuda:0", seed=0,
                 env_offset=0, auto_reset=True, reference_reset_obs=True, params=None, collect_qp_stats=False):
        """scenario: 'PredatorCapturePrey' | 'Warehouse' | 'MaterialTransport' | 'Simple' | 'ArcticTransport'
        (wrapper.py:12-16).
        config_path / overrides: the reference's scenario YAML (same keys) and a dict of overrides.
        seed: key of the device reset sampler (one Philox stream per (global env, episode)); None draws one
            from os.urandom, the reference's `seed: -1` = "do not seed" (PredatorCapturePrey.py:27-28).
        env_offset: global index of env 0 of this shard (RNG streams are keyed by global index).
        auto_reset: finished envs are reset inside the step launch.
        reference_reset_obs: reset() returns zeros like the reference (PredatorCapturePrey.py:136);
            False returns the observation of the fresh state (get_obs()).
        params: a ready RgScenarioParams (e.g. received by broadcast) instead of a config."""
        self.lib = _lib.load()
        self.scenario = scenario
        self.cfg = None
        if params is None:
            self.cfg = load_config(scenario, config_path, overrides)
            params = make_params(scenario, self.cfg)
        self.params = params
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RobogymError("VecRobotariumEnv runs on an AMD GPU only (device='cuda:N'); no CPU path exists")
        if not torch.cuda.is_available():
            raise _lib.RobogymError("no HIP device visible to torch; marbler_amd has no CPU fallback")
        self.E = int(num_envs)
        self.N = int(params.n_agents)
        self.D = int(params.obs_dim)
        self.P = int(params.num_prey)
        if seed is None:
            import os
            seed = int.from_bytes(os.urandom(8), "little")
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.env_offset = int(env_offset)
        self.auto_reset = bool(auto_reset)
        self.reference_reset_obs = bool(reference_reset_obs)
        E, N, D, P = self.E, self.N, self.D, max(self.P, 1)
        dev = self.device
        f32, i32, u8 = torch.float32, torch.int32, torch.uint8
        zeros = self._alloc      # every device array of this object comes from here (tests carve them out of a guarded slab)
        # ---- state (rg_state)
        self.poses = zeros((E, 3, N,), f32)
        self.carry_dist = zeros((E, N,), f32)
        self.episode_steps = zeros((E,), i32)
        self.reset_count = zeros((E,), i32)
        self.prey_loc = zeros((E, P, 2,), f32)
        self.prey_sensed = zeros((E, P,), u8)
        self.prey_captured = zeros((E, P,), u8)
        self.loaded = zeros((E, N,), u8)
        self.load = zeros((E, N,), i32)
        self.zone_load = zeros((E, 2,), i32)
        self.messages = zeros((E, 4,), i32)
        self.grid = zeros((E, 96,), u8)
        self.goal_col = zeros((E,), i32, fill=1)
        self.pixel_type = zeros((E, N,), u8)
        self.reached_goal = zeros((E, N,), u8)
        # ---- rollout statistics (misc.py:151-206 accumulators, on device)
        self.ep_return = zeros((E,), f32)
        self.done_return_sum = zeros((E,), f32)
        self.done_count = zeros((E,), i32)
        self.done_steps_sum = zeros((E,), i32)
        # ---- scratch of the lane-group kernel: every env's NEXT initial state, drawn ahead of time (derived data, not
        # part of a snapshot).  Large batches run the thread-per-env kernel, which has no use for it.
        self.next_init = self.next_episode = None
        if E <= 65536:
            stride = self.lib.rg_next_init_stride(C.byref(params))
            if stride <= 0:
                raise _lib.RobogymError("rg_next_init_stride failed: " + self.lib.rg_last_error().decode())
            self.next_init = zeros((E, stride,), f32)
            self.next_episode = zeros((E,), i32, fill=-1)
        self._alloc_outputs()
        self.qp_sweeps = zeros((E,), i32) if collect_qp_stats else None

        if dev.index is None:
            self.device = dev = torch.device("cuda", torch.cuda.current_device())
        self._stream = torch.cuda.current_stream(dev)
        self._stream_ptr = self._stream.cuda_stream
        self._h = self.lib.rg_create(C.byref(params), self.E, self.env_offset, dev.index,
                                     C.c_void_p(self._stream_ptr))
        if not self._h:
            raise _lib.RobogymError("rg_create failed: " + self.lib.rg_last_error().decode())
        st = _lib.RgState(*(t.data_ptr() for t in (
            self.poses, self.carry_dist, self.episode_steps, self.reset_count, self.prey_loc, self.prey_sensed,
            self.prey_captured, self.loaded, self.load, self.zone_load, self.messages, self.grid, self.goal_col,
            self.pixel_type, self.reached_goal, self.ep_return,
            self.done_return_sum, self.done_count, self.done_steps_sum)),
            self.next_init.data_ptr() if self.next_init is not None else None,
            self.next_episode.data_ptr() if self.next_episode is not None else None)
        _lib.check(self.lib.rg_bind_state(self._h, C.byref(st)), "rg_bind_state")
        self._io = _lib.RgStepIO(self.obs.data_ptr(), self.reward.data_ptr(), self.done_u8.data_ptr(),
                                 self.dist_travelled.data_ptr(), self.violation.data_ptr(),
                                 self.remaining.data_ptr(),
                                 self.qp_sweeps.data_ptr() if self.qp_sweeps is not None else None)
        self._io_ref = C.byref(self._io)
        self._actions_i32 = zeros((E, N), i32)
        self.time_limit = 0
        self.elapsed = self.truncated = self.ended = self.reward_sum = None

    def _alloc_outputs(self):
        """Step outputs (rg_step_io): views of ONE allocation, so a host-side consumer (the single-env Wrapper)
        fetches everything a step returns with one device-to-host copy."""
        E, N, D = self.E, self.N, self.D
        f32, i32, u8 = torch.float32, torch.int32, torch.uint8
        sizes = [("obs", E * N * D * 4), ("reward", E * N * 4), ("dist_travelled", E * N * 4), ("remaining", E * 4),
                 ("done_u8", E), ("violation", E)]
        offs, total = {}, 0
        for name, nbytes in sizes:
            offs[name] = total
            total += (nbytes + 15) // 16 * 16
        self._out_arena = self._alloc((total,), u8)
        self._out_offsets = offs

        def view(name, nbytes, dtype, shape):
            return self._out_arena[offs[name]:offs[name] + nbytes].view(dtype).view(shape)

        self.obs = view("obs", E * N * D * 4, f32, (E, N, D))
        self.reward = view("reward", E * N * 4, f32, (E, N))
        self.dist_travelled = view("dist_travelled", E * N * 4, f32, (E, N))
        self.remaining = view("remaining", E * 4, i32, (E,))
        self.remaining.fill_(-1)
        self.done_u8 = view("done_u8", E, u8, (E,))
        self.done = self.done_u8.view(torch.bool)   # the same bytes (the kernel writes 0 / 1): no conversion launch per step
        self.violation = view("violation", E, u8, (E,))

    def _alloc(self, shape, dtype, fill=0):
        """One device array (state, output or scratch).  The library never allocates: torch owns the memory."""
        if fill == 0:
            return torch.zeros(*shape, dtype=dtype, device=self.device)
        return torch.full(tuple(shape), fill, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ reference surface
    @property
    def n_agents(self):
        return self.N

    @property
    def num_envs(self):
        return self.E

    def _sync_stream(self):
        """Launches go to torch's CURRENT stream of the env's device (so they are ordered with the torch ops
        around them, also under `with torch.cuda.stream(s)`); the C side selects the device itself."""
        ptr = torch.cuda.current_stream(self.device).cuda_stream
        if ptr != self._stream_ptr:
            _lib.check(self.lib.rg_set_stream(self._h, C.c_void_p(ptr)), "rg_set_stream")
            self._stream_ptr = ptr

    def reset(self, mask=None, book_episode=False, reference_rng=None, py_random=None):
        """scenario.reset() for all envs (mask=None) or those with mask != 0.  Returns obs [E,N,D]:
        zeros where the reference would (reference_reset_obs), else the fresh observation.
        The running episode return of a reset env restarts at zero; book_episode=True first counts the
        abandoned episode in done_return_sum / done_count / done_steps_sum (an episode cut short by a
        time limit outside the scenario, e.g. gym's TimeLimit).
        reference_rng: parity mode (row a17) -- a seeded `np.random.RandomState` (or the `np.random` module)
        from which the initial conditions are drawn on the host exactly as the reference draws them
        (marbler_amd/reference_reset.py), env by env in index order, and uploaded; the device sampler's
        Philox streams stay the throughput path."""
        self._sync_stream()
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = mask.data_ptr()
        _lib.check(self.lib.rg_reset(self._h, mptr, self.seed, _lib.RESET_BOOK_EPISODE if book_episode else 0), "rg_reset")
        if self.elapsed is not None:                 # a reset env's TimeLimit counter restarts
            if mask is None:
                self.elapsed.zero_()
            else:
                self.elapsed.masked_fill_(mask.bool(), 0)
        if reference_rng is not None:
            self._upload_reference_reset(mask, reference_rng, py_random)
        if self.reference_reset_obs:
            if mask is None:
                self.obs.zero_()
            else:
                self.obs.masked_fill_(mask.bool()[:, None, None], 0.0)   # no host round trip (boolean indexing would sync)
            return self.obs
        return self.get_obs()

    def _upload_reference_reset(self, mask, rng, py_random):
        import numpy as np
        from .reference_reset import draw_reset
        if self.cfg is None:
            raise ValueError("reference_rng needs the scenario config (construct the env from a config, not from params)")
        idx = list(range(self.E)) if mask is None else [int(i) for i in torch.nonzero(mask).flatten().cpu()]
        if not idx:
            return
        draws = [draw_reset(self.scenario, self.cfg, rng, py_random) for _ in idx]
        ix = torch.as_tensor(idx, device=self.device)

        def put(dst, key, dtype):
            arr = np.stack([np.asarray(d[key]) for d in draws]).astype(dtype)
            dst[ix] = torch.as_tensor(arr, device=self.device).reshape((len(idx),) + tuple(dst.shape[1:]))

        put(self.poses, "poses", np.float32)
        if "prey_loc" in draws[0]:
            put(self.prey_loc, "prey_loc", np.float32)
        if "zone_load" in draws[0]:
            put(self.zone_load, "zone_load", np.int32)
        if "grid" in draws[0]:
            put(self.grid, "grid", np.uint8)
            self.goal_col[ix] = torch.as_tensor([d["goal_col"] for d in draws], dtype=torch.int32, device=self.device)

    def enable_time_limit(self, time_limit):
        """gym's TimeLimit and EPyMARL's gymma reductions inside the step launch (rg_step_io's gymma block): every step()
        then also fills `reward_sum` [E] f32 (sum over agents), `truncated` [E] bool (the limit, not the scenario, ended
        the episode), `ended` [E] bool (done | truncated); an ended env restarts in the same launch (auto_reset) and a
        truncated episode is booked in the episode statistics.  `elapsed` [E] i32 is TimeLimit's counter (state)."""
        E, dev = self.E, self.device
        self.time_limit = int(time_limit)
        if self.time_limit < 1:
            raise ValueError("time_limit must be > 0")
        self.elapsed = self._alloc((E,), torch.int32)
        # the three per-step outputs of the block are views of ONE allocation [reward_sum f32 | ended u8 | truncated u8], so
        # that a consumer who must keep a step's outputs copies them with one launch (gymma_outputs_copy)
        pad = (E + 15) // 16 * 16
        self._gymma_arena = self._alloc((4 * pad + 2 * pad,), torch.uint8)
        self._gymma_pad = pad
        self.reward_sum, self._ended_u8, self._trunc_u8 = self._gymma_views(self._gymma_arena)
        self.truncated, self.ended = self._trunc_u8.view(torch.bool), self._ended_u8.view(torch.bool)
        self._io.elapsed, self._io.truncated = self.elapsed.data_ptr(), self._trunc_u8.data_ptr()
        self._io.ended, self._io.reward_sum = self._ended_u8.data_ptr(), self.reward_sum.data_ptr()
        self._io.time_limit = self.time_limit
        # a second argument block for step_into(): the same buffers, three pointers replaced per call, zeros for ended envs
        self._io_into = _lib.RgStepIO.from_buffer_copy(self._io)
        self._io_into.zero_obs_on_end = 1
        self._io_into_ref = C.byref(self._io_into)

    def _gymma_views(self, arena):
        E, pad = self.E, self._gymma_pad
        return arena[:4 * E].view(torch.float32), arena[4 * pad:4 * pad + E], arena[5 * pad:5 * pad + E]

    def step_into(self, actions_ptr, obs_ptr, reward_sum_ptr, ended_ptr):
        """One rg_step whose gymma-shaped outputs go straight into a consumer's buffers (a trainer's time-major batch): the
        observation [E,N,D] -- with the rows of an env that ends in this step written as ZEROS, the reset observation a gymma
        user sees next (rg_step_io.zero_obs_on_end) --, the summed reward [E] f32 and the episode-end flags [E] u8.  Device
        pointers; needs enable_time_limit().  Everything else a step returns (reward per agent, done, dist_travelled,
        violation, remaining, truncated) lands in the env's own buffers as usual; `self.obs` is NOT written."""
        io = self._io_into
        io.obs, io.reward_sum, io.ended = obs_ptr, reward_sum_ptr, ended_ptr
        return self.lib.rg_step(self._h, actions_ptr, self._io_into_ref, 1 if self.auto_reset else 0, self.seed)

    def gymma_outputs_copy(self):
        """(reward_sum [E] f32, ended [E] bool, truncated [E] bool) of the last step as FRESH tensors: one copy launch."""
        r, e, t = self._gymma_views(self._gymma_arena.clone())
        return r, e.view(torch.bool), t.view(torch.bool)

    def step(self, actions):
        """actions: int tensor [E,N] on the device (int32 is used as is; other int dtypes are
        converted).  Returns (obs, reward, done, info) as views of the env's output buffers --
        they are overwritten by the next step."""
        self._sync_stream()
        if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
            self._actions_i32.copy_(actions.reshape(self.E, self.N))
            actions = self._actions_i32
        rc = self.lib.rg_step(self._h, actions.data_ptr(), self._io_ref, 1 if self.auto_reset else 0, self.seed)
        if rc != 0:
            _lib.check(rc, "rg_step")
        return self.obs, self.reward, self.done, self.info

    @property
    def info(self):
        return {"dist_travelled": self.dist_travelled, "violation": self.violation, "remaining": self.remaining}

    def host_step(self, actions):
        """One step of a single-env object (E == 1) for host-side consumers (the reference-typed Wrapper,
        EPyMARL's gymma shape): the actions go up through one pinned buffer, everything the step returns
        comes down with ONE copy of the output allocation.  Returns NumPy views of that host copy
        (obs [N,D] f32, reward [N] f32, done bool, violation int, remaining int, dist_travelled [N] f32),
        valid until the next call."""
        if self.E != 1:
            raise ValueError("host_step is the single-env path (num_envs == 1)")
        import numpy as np
        self._sync_stream()     # the launch, the upload and the download below all go to torch's CURRENT stream
        if getattr(self, "_hs", None) is None:
            o, N, D = self._out_offsets, self.N, self.D
            host = torch.zeros_like(self._out_arena, device="cpu").pin_memory()
            n = host.numpy()
            self._hs = {"act_host": torch.zeros(1, N, dtype=torch.int32).pin_memory(),
                        "act_dev": torch.zeros(1, N, dtype=torch.int32, device=self.device), "host": host,
                        "obs": n[o["obs"]:o["obs"] + N * D * 4].view(np.float32).reshape(N, D),
                        "reward": n[o["reward"]:o["reward"] + N * 4].view(np.float32),
                        "dist": n[o["dist_travelled"]:o["dist_travelled"] + N * 4].view(np.float32),
                        "remaining": n[o["remaining"]:o["remaining"] + 4].view(np.int32),
                        "done": n[o["done_u8"]:o["done_u8"] + 1], "viol": n[o["violation"]:o["violation"] + 1]}
        hs = self._hs
        hs["act_host"].numpy()[0, :] = np.asarray(actions, dtype=np.int32).reshape(self.N)
        hs["act_dev"].copy_(hs["act_host"], non_blocking=True)
        rc = self.step_raw(hs["act_dev"].data_ptr())
        if rc != 0:
            _lib.check(rc, "rg_step")
        hs["host"].copy_(self._out_arena, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return hs["obs"], hs["reward"], bool(hs["done"][0]), int(hs["viol"][0]), int(hs["remaining"][0]), hs["dist"]

    @property
    def step_kernel(self):
        """'group' (a lane group per env) or 'tpe' (one lane per env): the kernel this env's handle launches
        (robogym_capi.hip: tpe_min_envs; RG_STEP_KERNEL forces one).  Results are bit-identical either way."""
        return "tpe" if self.lib.rg_step_kernel(self._h) == 1 else "group"

    def step_raw(self, actions_ptr):
        """Hot-loop entry: one rg_step on a pre-validated device pointer to int32 [E,N]; results are
        in self.obs / reward / done_u8 / dist_travelled / violation / remaining.  Launches on the stream of
        the last step() / reset() / set_stream() call (no per-call stream lookup)."""
        return self.lib.rg_step(self._h, actions_ptr, self._io_ref, 1 if self.auto_reset else 0, self.seed)

    def set_stream(self, stream=None):
        """Send later launches to `stream` (a torch.cuda.Stream; None = the current stream of the env's
        device).  With a capturing stream the launches are recorded into the hipGraph being captured."""
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        _lib.check(self.lib.rg_set_stream(self._h, C.c_void_p(stream.cuda_stream)), "rg_set_stream")
        self._stream = stream
        self._stream_ptr = stream.cuda_stream

    def rollout(self, actions, out=None):
        """K env steps in one launch for an action sequence known up front (random-policy rollouts,
        replayed logs, open-loop plans): actions int32 [K,E,N] on the device.  Returns a dict of
        [K,...] tensors -- obs [K,E,N,D], reward [K,E,N], done [K,E] (uint8), dist_travelled [K,E,N],
        violation [K,E], remaining [K,E] -- holding exactly what K calls of step() would have
        returned (bit-identical; auto-reset as configured).  `out`: a dict from a previous call to
        reuse its buffers.  The single-step buffers (self.obs, ...) are left untouched."""
        if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        self._sync_stream()
        K = int(actions.shape[0])
        if tuple(actions.shape) != (K, self.E, self.N):
            raise ValueError(f"actions must be [K,{self.E},{self.N}], got {tuple(actions.shape)}")
        if out is None or out["obs"].shape[0] != K:
            f32, new = torch.float32, self._alloc
            out = {"obs": new((K, self.E, self.N, self.D), f32),
                   "reward": new((K, self.E, self.N), f32),
                   "done": new((K, self.E), torch.uint8),
                   "dist_travelled": new((K, self.E, self.N), f32),
                   "violation": new((K, self.E), torch.uint8),
                   "remaining": new((K, self.E), torch.int32)}
            if self.qp_sweeps is not None:
                out["qp_sweeps"] = new((K, self.E), torch.int32)
            out["_io"] = _lib.RgStepIO(out["obs"].data_ptr(), out["reward"].data_ptr(), out["done"].data_ptr(),
                                       out["dist_travelled"].data_ptr(), out["violation"].data_ptr(),
                                       out["remaining"].data_ptr(),
                                       out["qp_sweeps"].data_ptr() if "qp_sweeps" in out else None)
        rc = self.lib.rg_rollout(self._h, actions.data_ptr(), K, C.byref(out["_io"]), 1 if self.auto_reset else 0,
                                 self.seed)
        if rc != 0:
            _lib.check(rc, "rg_rollout")
        return out

    def get_obs(self, out=None):
        """Observation of the current state without stepping (gymma's get_obs())."""
        self._sync_stream()
        out = self.obs if out is None else out
        _lib.check(self.lib.rg_get_obs(self._h, out.data_ptr()), "rg_get_obs")
        return out

    # ------------------------------------------------------------------ state access (parity / checkpoints)
    # everything a later step depends on: the scenario state, the RNG stream position (reset_count) and the
    # rollout-statistics accumulators -- a restored snapshot continues bit-identically, statistics included
    STATE_KEYS = ("poses", "carry_dist", "episode_steps", "reset_count", "prey_loc", "prey_sensed", "prey_captured",
                  "loaded", "load", "zone_load", "messages", "grid", "goal_col", "pixel_type", "reached_goal",
                  "ep_return", "done_return_sum", "done_count", "done_steps_sum")

    def state_dict(self):
        """Snapshot of the env state (cloned tensors) + the sampler key, loadable with load_state_dict()."""
        sd = {k: getattr(self, k).clone() for k in self.STATE_KEYS}
        if self.elapsed is not None:
            sd["elapsed"] = self.elapsed.clone()     # gym TimeLimit's counter (enable_time_limit)
        sd["seed"] = torch.tensor([self.seed & 0xFFFFFFFF, self.seed >> 32], dtype=torch.int64)
        return sd

    def load_state_dict(self, sd):
        """Restore a snapshot.  The TimeLimit counter (`elapsed`) travels with snapshots of time-limited envs: a
        time-limited env REQUIRES it (a stale counter would move the next truncation), an env without a time limit
        refuses a snapshot that carries one (it would be dropped silently)."""
        if self.elapsed is not None and "elapsed" not in sd:
            raise KeyError("snapshot has no 'elapsed' (gym TimeLimit counter) but this env has a time limit: take the "
                           "snapshot from a time-limited env, or add sd['elapsed'] (zeros = every episode just started)")
        if self.elapsed is None and "elapsed" in sd:
            raise KeyError("snapshot carries 'elapsed' (gym TimeLimit counter) but this env has no time limit: call "
                           "enable_time_limit() first, or drop the key")
        unknown = [k for k in sd if k != "seed" and k != "elapsed" and k not in self.STATE_KEYS]
        if unknown:
            raise KeyError(f"unknown snapshot keys {unknown}")
        if self.next_episode is not None:
            self.next_episode.fill_(-1)     # drawn-ahead initial states belong to the state that is being replaced
        for k, v in sd.items():
            if k == "seed":
                self.seed = int(v[0]) | (int(v[1]) << 32)
                continue
            getattr(self, k).copy_(torch.as_tensor(v).to(getattr(self, k).dtype).reshape(getattr(self, k).shape))

    def episode_stats(self):
        """(sum of finished-episode returns, number of finished episodes, total steps in them) as
        0-dim device tensors -- the quantities run_env prints per run (misc.py:219-221)."""
        return self.done_return_sum.sum(), self.done_count.sum(), self.done_steps_sum.sum()

    def close(self):
        if getattr(self, "_h", None):
            self.lib.rg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
