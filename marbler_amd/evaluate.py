"""Batched greedy evaluation of EPyMARL actors on the device.

The reference evaluates a trained model with `python -m robotarium_gym.main` -> `run_env`
(utilities/misc.py:134-221): one env, one step at a time, actor forward on the CPU.  Here the
same loop runs for E envs at once: observations never leave the GPU, the actor
(utilities/rnn_agent.py:5-29 `RNNAgent`: fc1 -> ReLU -> GRUCell -> fc2, or the non-shared
`RNNNSAgent` of rnn_ns_agent.py:5-36 with one such network per agent) is evaluated for all
E x N agents with a handful of batched GEMMs (torch -> rocBLAS/hipBLASLt: plain library GEMMs),
actions are the arg-max Q (misc.py:170), and the episode statistics accumulate in the env's
on-device counters.  `load_actor` reads the reference's model zoo (`scenarios/*/models/*.th`
state dicts + sacred `.json` configs, misc.py:65-91) unchanged.
"""
import json

import torch


DEFAULT_PACK_GRU = "f16x2"   # what pack_gru=True means


class BatchedActor(object):
    """RNNAgent / RNNNSAgent weights stacked as [A, ...] with A = 1 (shared) or N (one per agent)."""

    def __init__(self, state_dict, n_agents, use_rnn=True, device="cpu", pack_gru=True):
        keys = list(state_dict.keys())
        self.non_shared = keys[0].startswith("agents.")
        self.use_rnn = bool(use_rnn)
        self.n_agents = int(n_agents)
        # fused kernel: how the two GRU matrices are handed over -- True / "f16x2": two binary16 planes (three plane products per float32
        # product on the binary16 matrix cores, error at the level of a float32 GEMM's own roundings; activations above 65 504
        # saturate); "bf16x3": three bfloat16 planes (six plane products, float32's exponent range, error below a float32 GEMM's
        # own; 1.2 x the launch time); "f32": float32 in the kernel's streaming order (f32-input MFMA, 16 x slower per product);
        # False: torch's layout as it is (f32-input MFMA)
        self.pack_gru = {True: DEFAULT_PACK_GRU, False: None, None: None}.get(pack_gru, pack_gru)
        if self.pack_gru not in (None, "bf16x3", "f16x2", "f32"):
            raise ValueError("pack_gru must be True, 'f16x2', 'bf16x3', 'f32' or False")

        def stack(name):
            if self.non_shared:
                return torch.stack([state_dict[f"agents.{i}.{name}"] for i in range(self.n_agents)]).float().to(device)
            return state_dict[name].float().unsqueeze(0).to(device)

        self.w1, self.b1 = stack("fc1.weight"), stack("fc1.bias")          # [A,H,I], [A,H]
        self.w2, self.b2 = stack("fc2.weight"), stack("fc2.bias")          # [A,O,H], [A,O]
        if self.use_rnn:
            self.wih, self.whh = stack("rnn.weight_ih"), stack("rnn.weight_hh")   # [A,3H,H]
            self.bih, self.bhh = stack("rnn.bias_ih"), stack("rnn.bias_hh")
        else:
            self.wr, self.br = stack("rnn.weight"), stack("rnn.bias")
        self.hidden_dim = self.w1.shape[1]
        self.input_dim = self.w1.shape[2]
        self.n_actions = self.w2.shape[1]

    def init_hidden(self, num_envs):
        return torch.zeros(num_envs, self.n_agents, self.hidden_dim, device=self.w1.device)

    def _lin(self, x, w, b):
        # x [E,N,I]; w [A,O,I]; b [A,O] -> [E,N,O]
        if w.shape[0] == 1:
            return torch.matmul(x, w[0].t()) + b[0]
        return torch.einsum("eni,noi->eno", x, w) + b

    def forward(self, inputs, hidden):
        """inputs [E,N,input_dim], hidden [E,N,H] -> (q [E,N,n_actions], new hidden [E,N,H])."""
        x = torch.relu(self._lin(inputs, self.w1, self.b1))
        if self.use_rnn:   # torch.nn.GRUCell: r, z, n gates in that order
            gi = self._lin(x, self.wih, self.bih)
            gh = self._lin(hidden, self.whh, self.bhh)
            H = self.hidden_dim
            r = torch.sigmoid(gi[..., :H] + gh[..., :H])
            z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
            n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
            h = (1.0 - z) * n + z * hidden
        else:
            h = torch.relu(self._lin(x, self.wr, self.br))
        return self._lin(h, self.w2, self.b2), h

    # ------------------------------------------------------------------ fused HIP kernel (csrc/actor_mfma.hip)
    def fused_supported(self):
        return (self.w1.is_cuda and self.hidden_dim in (64, 128) and self.n_actions <= 32 and self.input_dim <= 64)

    def _weights_struct(self):
        import ctypes as C
        from . import _lib
        if getattr(self, "_ws", None) is None:
            t = {k: getattr(self, k).contiguous() for k in ("w1", "b1", "w2", "b2")}
            packed = 0
            if self.use_rnn:
                t.update({k: getattr(self, k).contiguous() for k in ("wih", "bih", "whh", "bhh")})
            if self.use_rnn and self.pack_gru:
                lib = _lib.load()   # the two GRU matrices once into the kernel's streaming order
                stream = C.c_void_p(torch.cuda.current_stream(self.w1.device).cuda_stream)
                for k in ("wih", "whh"):
                    if self.pack_gru == "bf16x3":
                        dst = torch.empty(t[k].numel() * 3, dtype=torch.int16, device=t[k].device)   # 6 bytes per weight
                        rc = lib.rg_actor_pack_gru_bf16x3(t[k].data_ptr(), self.w1.shape[0], self.hidden_dim, dst.data_ptr(), stream)
                    elif self.pack_gru == "f16x2":
                        dst = torch.empty(t[k].numel() * 2, dtype=torch.int16, device=t[k].device)   # 4 bytes per weight
                        rc = lib.rg_actor_pack_gru_f16x2(t[k].data_ptr(), self.w1.shape[0], self.hidden_dim, dst.data_ptr(), stream)
                    else:
                        dst = torch.empty_like(t[k])
                        rc = lib.rg_actor_pack_gru(t[k].data_ptr(), self.w1.shape[0], self.hidden_dim, dst.data_ptr(), stream)
                    if rc != 0:
                        raise _lib.RobogymError("rg_actor_pack_gru: " + lib.rg_actor_last_error().decode())
                    t[k] = dst
                packed = {"bf16x3": 2, "f16x2": 3}.get(self.pack_gru, 1)
                # a first launch on ANOTHER stream must not overtake the two pack kernels (forward_fused waits on this once)
                self._pack_done = torch.cuda.Event()
                self._pack_done.record(torch.cuda.current_stream(self.w1.device))
            if not self.use_rnn:
                t.update({"wih": self.wr.contiguous(), "bih": self.br.contiguous()})
            ptr = lambda k: t[k].data_ptr() if k in t else None  # noqa: E731
            self._ws_tensors = t   # keep the contiguous copies alive
            self._ws = _lib.RgActorWeights(ptr("w1"), ptr("b1"), ptr("wih"), ptr("bih"), ptr("whh"), ptr("bhh"),
                                           ptr("w2"), ptr("b2"), self.w1.shape[0], self.input_dim, self.hidden_dim,
                                           self.n_actions, 1 if self.use_rnn else 0, packed)
        return self._ws

    def forward_fused(self, obs, hidden, append_agent_id=True, restart=None, q_out=None, actions_out=None, stream=None,
                      explore_u=None, epsilon=0.0):
        """One actor step for all E x N agents in one launch (rg_actor_forward: the matrix cores; GRU on bfloat16 planes unless pack_gru says otherwise).
        obs [E,N,D] f32; hidden [E,N,H] f32 updated IN PLACE; restart [E] uint8 (nonzero = start that
        env's hidden state from zero) or None; stream: a torch.cuda.Stream (default: the device's current stream).
        explore_u [E,N] f32 uniforms in [0, 1) with epsilon > 0: epsilon-greedy actions (rg_actor_forward_explore; the rule is
        `explore_select` below).  Returns (q [E,N,A], actions [E,N] int32)."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        E, N, D = obs.shape
        if q_out is None:
            q_out = torch.empty(E, N, self.n_actions, device=obs.device)
        if actions_out is None:
            actions_out = torch.empty(E, N, dtype=torch.int32, device=obs.device)
        if stream is None:
            stream = torch.cuda.current_stream(obs.device)
        ws = self._weights_struct()
        ev = getattr(self, "_pack_done", None)
        if ev is not None:   # the weights were packed (once) on the then-current stream: order this stream behind them
            if ev.query():
                self._pack_done = None
            else:
                stream.wait_event(ev)
        if explore_u is not None:
            if explore_u.shape != (E, N) or explore_u.dtype != torch.float32 or not explore_u.is_contiguous():
                raise ValueError("explore_u must be a contiguous float32 [E, N] tensor")
            rc = lib.rg_actor_forward_explore(C.byref(ws), E, N, obs.data_ptr(), D, 1 if append_agent_id else 0,
                                              restart.data_ptr() if restart is not None else None, hidden.data_ptr(),
                                              q_out.data_ptr(), actions_out.data_ptr(), explore_u.data_ptr(), float(epsilon),
                                              C.c_void_p(stream.cuda_stream))
        else:
            rc = lib.rg_actor_forward(C.byref(ws), E, N, obs.data_ptr(), D, 1 if append_agent_id else 0,
                                      restart.data_ptr() if restart is not None else None, hidden.data_ptr(),
                                      q_out.data_ptr(), actions_out.data_ptr(), C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise _lib.RobogymError("rg_actor_forward: " + lib.rg_actor_last_error().decode())
        return q_out, actions_out


def explore_select(greedy, u, epsilon, n_actions, out=None):
    """The epsilon-greedy rule of rg_actor_forward_explore in torch ops (the composed runner path and the tests use it): with
    k = int(u * (n_actions / epsilon)) in float32, the action is k where k < n_actions (u < epsilon; k uniform), else greedy."""
    import numpy as np
    scale = float(np.float32(n_actions) / np.float32(epsilon))
    k = (u * torch.tensor(scale, dtype=torch.float32, device=u.device)).to(torch.int32)
    return torch.where(k < n_actions, k, greedy, out=out)


def load_actor(model_file, model_config, n_agents, device="cuda:0"):
    """model_file: a `.th` state dict of the reference's model zoo; model_config: its sacred `.json`
    (path or dict).  Returns (BatchedActor, config dict) -- misc.py:65-91 without the env part."""
    cfg = model_config if isinstance(model_config, dict) else json.load(open(model_config))
    sd = torch.load(model_file, map_location="cpu")
    return BatchedActor(sd, n_agents, use_rnn=cfg.get("use_rnn", True), device=device), cfg


@torch.no_grad()
def run_eval(env, actor, steps, obs_agent_id=True, use_graph=False, fused=None):
    """Greedy rollout of `actor` on a VecRobotariumEnv (auto_reset on) for `steps` env steps.
    Mirrors run_env's per-step body (misc.py:160-172): optional one-hot agent id appended to the
    observation, actor forward, arg-max, env.step; hidden states restart at zero with each episode.
    Returns the statistics run_env prints, from the env's on-device accumulators.

    use_graph: record one iteration (a dozen small GEMM / elementwise launches plus the env step)
    into a hipGraph on a side stream and replay it `steps` times -- the loop is launch-bound, the
    replay is one launch per step.  Same arithmetic, same results.
    fused (default: when the actor's shape allows): the whole policy step in one launch of the MFMA
    kernel (csrc/actor_mfma.hip) -- an iteration is then three launches (actor, env step, distance
    sum); float32 like the torch path, sums in a different order (action values agree to 1e-5)."""
    E, N = env.E, env.N
    dev = env.device
    eye = torch.eye(N, device=dev).unsqueeze(0).expand(E, N, N)
    in_dim = env.D + (N if obs_agent_id else 0)
    if in_dim != actor.input_dim:
        raise ValueError(f"actor expects {actor.input_dim} inputs per agent, the env provides {in_dim}")
    # loop-carried state lives in fixed buffers, updated in place (what a graph replay needs)
    obs_in = env.reset().clone()
    hidden = actor.init_hidden(E)
    dist = torch.zeros(E, N, device=dev)
    actions = torch.zeros(E, N, dtype=torch.int32, device=dev)

    def body():
        inp = torch.cat([obs_in, eye], dim=2) if obs_agent_id else obs_in
        q, h = actor.forward(inp, hidden)
        actions.copy_(q.argmax(dim=2))
        obs, reward, done, info = env.step(actions)
        dist.add_(info["dist_travelled"])
        keep = (~done)[:, None, None]
        hidden.copy_(torch.where(keep, h, 0.0))
        obs_in.copy_(torch.where(keep, obs, 0.0))   # a finished env restarts from the reference's reset() observation (zeros)

    if fused is None:
        fused = actor.fused_supported()
    if fused:
        q_buf = torch.empty(E, N, actor.n_actions, device=dev)

        def body():  # noqa: F811 - the fused form of the loop body above
            # env.obs / env.done_u8 are the previous step's outputs (zeros after reset()): a finished env is
            # seen through a zero observation and a zero hidden state, inside the kernel
            actor.forward_fused(env.obs, hidden, append_agent_id=obs_agent_id, restart=env.done_u8, q_out=q_buf,
                                actions_out=actions)
            env.step(actions)
            dist.add_(env.dist_travelled)

    if not use_graph:
        for _ in range(steps):
            body()
    else:
        side = torch.cuda.Stream(device=dev)
        prev = env._stream
        side.wait_stream(torch.cuda.current_stream(dev))
        try:
            env.set_stream(side)
            with torch.cuda.stream(side):
                n_warm = min(steps, 3)   # library handles and workspaces exist before the capture
                for _ in range(n_warm):
                    body()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    body()
                for _ in range(steps - n_warm):
                    graph.replay()
            torch.cuda.current_stream(dev).wait_stream(side)
        finally:
            env.set_stream(prev)
        torch.cuda.synchronize(dev)
    ret_sum, episodes, ep_steps = env.episode_stats()
    n = max(int(episodes), 1)
    return {"episodes": int(episodes), "mean_return": float(ret_sum) / n, "mean_steps": float(ep_steps) / n,
            "mean_dist_per_step": float(dist.sum()) / (steps * E * N)}


def main(argv=None):
    """`python -m marbler_amd.evaluate --scenario PredatorCapturePrey --model-file qmix.th --model-config qmix.json`
    -- the batched counterpart of `python -m robotarium_gym.main --scenario X` (main.py / misc.py:93-221):
    loads a model of the reference's zoo, rolls it out greedily on --envs envs for --steps steps, prints
    the statistics run_env prints (mean return, mean episode length) as one JSON line."""
    import argparse
    from .params import load_config
    from .vec_env import VecRobotariumEnv
    ap = argparse.ArgumentParser(prog="python -m marbler_amd.evaluate")
    ap.add_argument("--scenario", required=True)
    ap.add_argument("--model-file", required=True, help="a .th state dict of the model zoo")
    ap.add_argument("--model-config", required=True, help="its sacred .json config")
    ap.add_argument("--config", default=None, help="scenario YAML (default: this package's copy of the reference's)")
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--torch-actor", action="store_true", help="evaluate the policy with torch ops instead of the fused kernel")
    args = ap.parse_args(argv)
    cfg = load_config(args.scenario, config_path=args.config)
    env = VecRobotariumEnv(args.scenario, args.envs, overrides=cfg, device=args.device, seed=args.seed)
    actor, mcfg = load_actor(args.model_file, args.model_config, env.N, device=args.device)
    out = run_eval(env, actor, args.steps, obs_agent_id=bool(mcfg.get("obs_agent_id", True)),
                   fused=False if args.torch_actor else None)
    out.update({"scenario": args.scenario, "envs": args.envs, "steps": args.steps})
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
