"""The reference's own trained policies, evaluated on this repo's engine -> tests/golden/ZOO_EVAL.json (+ zoo_eval_replay.npz).

    python tests/golden/make_zoo_eval.py [--episodes 1024] [--write]

Build container only (like make_golden.py): the 22 checkpoints stay where they are, under
/root/reference/robotarium_gym/scenarios/*/models/ ("the final models we evaluated in the paper",
robotarium_gym/README.md:63).  Each is loaded the way the reference loads it (utilities/misc.py:65-91: the sacred `.json`,
`torch.load` of the `.th` state dict, the reference's OWN `RNNAgent` / `RNNNSAgent` modules, utilities/rnn_agent.py:5-29,
rnn_ns_agent.py:5-36, imported from /root/reference) and rolled out greedily with `run_env`'s loop body (misc.py:155-185: agent
id appended when `obs_agent_id`, actor forward in float32 on the CPU, arg-max, step; reset() observation = zeros, hidden
state = zeros; an episode ends with done[0]) on the C oracle at the scenario's shipped configuration -- one episode per env,
`--episodes` envs, every variant from the SAME initial states (sampler twin, seed 2024) --

  * float32            : sim_spec_v0 in binary32 = the HIP kernels, bit for bit (tests/test_gpu_*.py);
  * float64_exact      : the same spec in float64 (barrier QP = exact projection);
  * float32_cvxopt_restated : the float32 tier with the barrier QP as the restated cvxopt iterate (`barrier_solver: cvxopt`,
    ipm_spec_v0) = the HIP kernels' interior-point mode, bit for bit (round 5);
  * float64_cvxopt_restated : float64 with the barrier QP as a restated cvxopt interior-point iterate at the reference's
                         tolerances (oracle_core.h barrier_qp_ipm) -- a STUDY of the unpinned solver layer, not a pin.

Why: rows a4-a10 of the hot path (rps + cvxopt) cannot be pinned in this image.  These are the
quantities the reference's paper tabulates for exactly these files (return, episode length, collisions / boundary exits), so a
reader who has the paper -- or the genuine stack -- can place this engine's unpinned layer against it.  No number from the
paper is quoted here: it is not in the reference tree.

zoo_eval_replay.npz: for three rows, the float32 tier's recorded actions of the first 64 episodes and their per-episode
statistics; tests/test_gpu_zoo_replay.py steps the HIP kernels with those actions and must reproduce the statistics bit
for bit (the checkpoints do not travel to the GPU box, actions do).
"""
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

SEED = 2024
VARIANTS = {"float32": (np.float32, {}), "float64_exact": (np.float64, {}), "float64_cvxopt_restated": (np.float64, {"qp_solver": "cvxopt_restated"}),
            # round 5: the float32 tier with `barrier_solver: cvxopt` = the HIP kernels' interior-point mode, bit for bit
            "float32_cvxopt_restated": (np.float32, {"barrier_solver": "cvxopt"})}
REPLAY_VARIANTS = {"float32": "", "float32_cvxopt_restated": "ipm__"}   # variant -> key prefix in zoo_eval_replay.npz
REPLAY_ROWS = ("PredatorCapturePrey/qmix", "Warehouse/vdn", "MaterialTransport/mappo")
REPLAY_ENVS = 64


def load_reference_actor(th_path, js_path, n_agents, n_actions):
    """misc.py:65-91 (`load_env_and_model` without the env): the reference's module classes, its state dict."""
    import torch
    from ref_harness import _install_stubs
    _install_stubs()
    from robotarium_gym.utilities.rnn_agent import RNNAgent
    from robotarium_gym.utilities.rnn_ns_agent import RNNNSAgent
    cfg = json.load(open(js_path))

    class Args(object):
        pass
    a = Args()
    a.hidden_dim, a.use_rnn, a.n_actions, a.n_agents = cfg["hidden_dim"], cfg.get("use_rnn", True), n_actions, n_agents
    weights = torch.load(th_path, map_location=torch.device("cpu"))
    input_dim = weights[list(weights.keys())[0]].shape[1]
    ns = cfg.get("agent") == "rnn_ns"
    model = (RNNNSAgent if ns else RNNAgent)(input_dim, a)
    model.load_state_dict(weights)
    model.eval()
    return model, cfg, ns, input_dim


def greedy_actions(model, ns, obs, hidden, obs_agent_id):
    """One policy step for E envs: obs [E,N,D] (any float dtype) -> (actions [E,N] int32, new hidden [E,N,H]); float32 on the CPU
    like `model(torch.Tensor(obs), torch.Tensor(hs))` (misc.py:160-170), arg-max with NumPy's first-maximum rule."""
    import torch
    E, N, D = obs.shape
    x = obs.astype(np.float32)
    if obs_agent_id:
        x = np.concatenate([x, np.broadcast_to(np.eye(N, dtype=np.float32), (E, N, N))], axis=2)
    with torch.no_grad():
        xt, ht = torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(hidden)
        if ns:   # rnn_ns_agent.py:22-27: agent i's own network on its own row
            qs, hs = [], []
            for i in range(N):
                q, h = model.agents[i](xt[:, i], ht[:, i])
                qs.append(q)
                hs.append(h)
            q, h = torch.stack(qs, dim=1), torch.stack(hs, dim=1)
        else:
            q, h = model(xt.reshape(E * N, -1), ht.reshape(E * N, -1))
            q, h = q.reshape(E, N, -1), h.reshape(E, N, -1)
    return np.argmax(q.numpy(), axis=2).astype(np.int32), h.numpy().copy()


def evaluate(scenario, cfg, rg_params, model, mcfg, ns, E, dtype, record=0):
    """One greedy episode per env on the C oracle.  -> (summary dict, replay dict or None)."""
    from helpers import oracle_reset, oracle_reset_params
    from oracle import c_oracle
    orc = c_oracle.OracleVecEnv(scenario, cfg, E, dtype=dtype)
    rp = oracle_reset_params(c_oracle, rg_params)
    for e in range(E):
        oracle_reset(c_oracle, orc, rp, SEED, e, 0)
    N, H = orc.N, mcfg["hidden_dim"]
    obs = np.zeros((E, N, orc.D), np.float32)          # reset() returns zeros (PredatorCapturePrey.py:136, warehouse.py:100, ...)
    hidden = np.zeros((E, N, H), np.float32)
    active = np.ones(E, bool)
    ret = np.zeros(E, np.float64)
    steps = np.zeros(E, np.int64)
    dist = np.zeros((E, N), np.float64)
    viol = np.zeros(E, np.int64)
    remaining = np.full(E, -1, np.int64)
    shared = bool(rg_params.shared_reward)
    acts = []
    for j in range(int(cfg["max_episode_steps"]) + 1):   # misc.py:155
        a, hidden = greedy_actions(model, ns, obs, hidden, bool(mcfg.get("obs_agent_id", True)))
        a[~active] = 0
        if record:
            acts.append(a[:record].astype(np.int8))
        o, r, d, info = orc.step(a, threads=8)
        rr = r.astype(np.float64)
        ret[active] += (rr[:, 0] if shared else rr.sum(axis=1))[active]
        dist[active] += info["dist_travelled"].astype(np.float64)[active]
        ended = active & (d != 0)
        steps[ended] = j + 1
        viol[ended] = info["violation"][ended]
        remaining[ended] = info["remaining"][ended]
        active &= ~ended
        obs = o.astype(np.float32)
        if not active.any():
            break
    assert not active.any(), "every scenario ends by max_episode_steps + 1"
    summary = {"episodes": int(E), "return_mean": float(ret.mean()), "return_std": float(ret.std()), "steps_mean": float(steps.mean()),
               "steps_std": float(steps.std()), "dist_mean_per_agent": [float(v) for v in dist.mean(axis=0)], "dist_std": float(dist.std()),
               "ended_by": {"collision": int((viol == 1).sum()), "boundary": int((viol == 2).sum()), "collision_and_boundary": int((viol == 3).sum()),
                            "scenario": int((viol == 0).sum())},
               "remaining_mean_of_reported": float(remaining[remaining >= 0].mean()) if (remaining >= 0).any() else None}
    replay = None
    if record:
        replay = {"actions": np.stack(acts), "return": ret[:record], "steps": steps[:record], "dist": dist[:record], "violation": viol[:record],
                  "remaining": remaining[:record]}
    return summary, replay


def main():
    import argparse
    import torch
    from marbler_amd.gymma import N_ACTIONS
    from marbler_amd.params import load_config, make_params
    from oracle import c_oracle
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=1024)
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--only", default=None, help="substring of Scenario/model")
    args = ap.parse_args()
    torch.set_num_threads(8)
    c_oracle.build_library()
    zoo = "/root/reference/robotarium_gym/scenarios"
    out = {"what": __doc__.split("\n\n")[0] + " -- see the docstring of tests/golden/make_zoo_eval.py", "generated_by": "python tests/golden/make_zoo_eval.py --write",
           "seed": SEED, "episodes_per_row": args.episodes, "torch": torch.__version__, "numpy": np.__version__, "models": {}}
    replays = {}
    for scenario in sorted(os.listdir(zoo)):
        mdir = os.path.join(zoo, scenario, "models")
        if scenario not in N_ACTIONS or not os.path.isdir(mdir):
            continue
        cfg = load_config(scenario)
        p = make_params(scenario, cfg)
        for js in sorted(glob.glob(os.path.join(mdir, "*.json"))):
            name = f"{scenario}/{os.path.basename(js)[:-5]}"
            th = js[:-5] + ".th"
            if not os.path.exists(th) or (args.only and args.only not in name):
                continue
            model, mcfg, ns, input_dim = load_reference_actor(th, js, p.n_agents, N_ACTIONS[scenario])
            assert input_dim == p.obs_dim + (p.n_agents if mcfg.get("obs_agent_id", True) else 0), name
            row = {"agent": mcfg.get("agent"), "hidden_dim": mcfg["hidden_dim"], "use_rnn": mcfg.get("use_rnn", True), "obs_agent_id": mcfg.get("obs_agent_id", True),
                   "algorithm": mcfg.get("name"), "variants": {}}
            for vname, (dtype, extra) in VARIANTS.items():
                rec = REPLAY_ENVS if (vname in REPLAY_VARIANTS and name in REPLAY_ROWS) else 0
                summary, replay = evaluate(scenario, dict(cfg, **extra), p, model, mcfg, ns, args.episodes, dtype, record=rec)
                row["variants"][vname] = summary
                if replay:
                    replays[(REPLAY_VARIANTS[vname], name)] = replay
            out["models"][name] = row
            v = row["variants"]
            print(name, {k: (round(v[k]["return_mean"], 3), round(v[k]["steps_mean"], 2), v[k]["ended_by"]["collision"], v[k]["ended_by"]["boundary"]) for k in v}, flush=True)
    if args.write:
        with open(os.path.join(HERE, "ZOO_EVAL.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
            f.write("\n")
        flat = {"rows": np.array(sorted({name for _, name in replays})), "seed": np.int64(SEED)}
        for (prefix, name), rp in replays.items():
            key = prefix + name.replace("/", "__")
            for k, v in rp.items():
                flat[f"{key}__{k}"] = v
        np.savez_compressed(os.path.join(HERE, "zoo_eval_replay.npz"), **flat)


if __name__ == "__main__":
    main()
