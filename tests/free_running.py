"""Free-running statistical equivalence of the float32 path (GPU kernels = float32 oracle, bit for bit) and the float64
oracle (= the reference's arithmetic, tests/test_oracle_golden.py): the evidence teacher-forced parity cannot give.

Both sides start from the same reset stream and get the same random actions; float32 rounding flips a threshold now and
then (a capture one step later, a collision one sub-step earlier), after which the two trajectories of that env part for
good -- expected.  What must agree is the DISTRIBUTION of what a trainer sees: episode lengths, returns, violation codes,
what is left at the end of an episode.  `compare` checks them within 3 sigma of the sampling error of two independent
runs (the runs are in fact strongly correlated, so this is generous in the test's favour only where it does not matter:
a systematic bias of the float32 path would show as many sigma).

Reference: utilities/misc.py:134-221 (run_env: the per-episode statistics an evaluation prints).
"""
import numpy as np


class Tally(object):
    """Per-step outputs -> per-episode statistics (the accumulators of run_env, misc.py:151-206)."""

    def __init__(self, E, max_len):
        self.ret = np.zeros(E, np.float64)
        self.len = np.zeros(E, np.int64)
        self.returns, self.lengths, self.remaining = [], [], []
        self.viol = np.zeros(4, np.int64)
        self.env_steps = 0
        self.max_len = max_len

    def add(self, reward_env, done, viol, remaining):
        """reward_env [E] (the episode reward increment: reward[0] when shared, else the sum over agents)."""
        self.ret += reward_env
        self.len += 1
        self.env_steps += len(done)
        self.viol += np.bincount(viol, minlength=4)[:4]
        idx = np.nonzero(done)[0]
        if len(idx):
            self.returns.append(self.ret[idx].copy())
            self.lengths.append(self.len[idx].copy())
            self.remaining.append(np.asarray(remaining)[idx].copy())
            self.ret[idx] = 0
            self.len[idx] = 0

    def summary(self):
        r = np.concatenate(self.returns) if self.returns else np.zeros(0)
        n = np.concatenate(self.lengths) if self.lengths else np.zeros(0, np.int64)
        m = np.concatenate(self.remaining) if self.remaining else np.zeros(0, np.int64)
        return {"episodes": int(len(r)), "env_steps": int(self.env_steps),
                "return_mean": float(r.mean()) if len(r) else 0.0, "return_var": float(r.var()) if len(r) else 0.0,
                "length_mean": float(n.mean()) if len(n) else 0.0, "length_var": float(n.var()) if len(n) else 0.0,
                "length_hist": np.bincount(n, minlength=self.max_len + 2)[:self.max_len + 2].tolist(),
                "violation_counts": self.viol.tolist(),
                "remaining_hist": np.bincount(np.clip(m, -1, 62) + 1, minlength=64).tolist()}


def compare(a, b, what="", nsigma=3.0, skip=()):
    """a, b: Tally.summary() of two runs.  Raises AssertionError naming the statistic that is off by more than nsigma
    sigma.  skip: substrings of statistic names left out (the solver-tolerance study compares collision counts on their own)."""
    def close(x, y, sigma, name):
        if any(k in name for k in skip):
            return
        assert abs(x - y) <= nsigma * sigma + 1e-9, f"{what}: {name} differs by {abs(x - y):.4g} > {nsigma} sigma = {nsigma * sigma:.4g} ({x} vs {y})"

    na, nb = a["episodes"], b["episodes"]
    assert na > 100 and nb > 100, f"{what}: too few episodes ({na}, {nb})"
    close(na, nb, np.sqrt(na + nb), "episode count")
    close(a["return_mean"], b["return_mean"], np.sqrt(a["return_var"] / na + b["return_var"] / nb), "mean return")
    close(a["length_mean"], b["length_mean"], np.sqrt(a["length_var"] / na + b["length_var"] / nb), "mean episode length")
    for code in range(4):
        x, y = a["violation_counts"][code], b["violation_counts"][code]
        close(x, y, np.sqrt(x + y) + 1.0, f"count of violation code {code}")
    for name in ("length_hist", "remaining_hist"):
        for i, (x, y) in enumerate(zip(a[name], b[name])):
            close(x, y, np.sqrt(x + y) + 1.0, f"{name}[{i}]")


def run_oracle(oracle_lib, scenario, cfg, rg_params, E, steps, n_act, dtype, seed, action_seed, threads=8):
    """The C oracle free-running with the reset twin (float-spec sampler; its values are exact in float64 too)."""
    from helpers import oracle_reset, oracle_reset_params
    orc = oracle_lib.OracleVecEnv(scenario, cfg, E, dtype=dtype)
    rp = oracle_reset_params(oracle_lib, rg_params)
    for e in range(E):
        oracle_reset(oracle_lib, orc, rp, seed, e, 0)
    episodes = np.zeros(E, np.int64)
    rng = np.random.RandomState(action_seed)
    tally = Tally(E, int(cfg["max_episode_steps"]) + 1)
    shared = bool(rg_params.shared_reward)
    for t in range(steps):
        a = rng.randint(0, n_act, size=(E, orc.N)).astype(np.int32)
        obs, rew, done, info = orc.step(a, threads=threads)
        tally.add(rew[:, 0].astype(np.float64) if shared else rew.astype(np.float64).sum(axis=1), done, info["violation"], info["remaining"])
        for e in np.nonzero(done)[0]:
            episodes[e] += 1
            oracle_reset(oracle_lib, orc, rp, seed, e, int(episodes[e]))
    return tally.summary()


def run_gpu(scenario, ov, E, steps, n_act, seed, action_seed):
    import torch
    from marbler_amd import VecRobotariumEnv
    env = VecRobotariumEnv(scenario, E, overrides=ov, seed=seed, auto_reset=True)
    env.reset()
    rng = np.random.RandomState(action_seed)
    tally = Tally(E, int(env.cfg["max_episode_steps"]) + 1)
    shared = bool(env.params.shared_reward)
    for t in range(steps):
        a = torch.as_tensor(rng.randint(0, n_act, size=(E, env.N)).astype(np.int32), device=env.device)
        obs, rew, done, info = env.step(a)
        r = rew.double().cpu().numpy()
        tally.add(r[:, 0] if shared else r.sum(axis=1), done.cpu().numpy().astype(np.uint8), info["violation"].cpu().numpy(),
                  info["remaining"].cpu().numpy())
    env.close()
    return tally.summary()


CASES = {   # name: (scenario, overrides, actions, envs, steps)
    "pcp_4096x5": ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 4096, 400),
    "warehouse_4096x8": ("Warehouse", {"n_agents": 8}, 5, 4096, 300),
    "mt_2048x6": ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20, 2048, 300),
}
SEED, ACTION_SEED = 2024, 77


if __name__ == "__main__":   # python tests/free_running.py --write : the float64 / float32 oracle statistics -> tests/golden/FREE_RUNNING_STATS.json
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    from marbler_amd.params import load_config, make_params
    from oracle import c_oracle
    c_oracle.build_library()
    out = {"what": "free-running random-policy rollouts from one reset stream and one action stream: per-episode statistics of the "
                   "float64 oracle (the reference's arithmetic) and of the float32 oracle (= the HIP kernels, bit for bit); "
                   "float64_cvxopt_restated: the same float64 oracle with the barrier QP solved by a restated cvxopt interior-point "
                   "method at the reference's tolerances instead of exactly (a study of what the unpinned solver difference is "
                   "worth, not a parity claim)",
           "seed": SEED, "action_seed": ACTION_SEED, "generated_by": "python tests/free_running.py --write", "cases": {}}
    for name, (scenario, ov, n_act, E, steps) in CASES.items():
        cfg = load_config(scenario, None, ov)
        p = make_params(scenario, cfg)
        f64 = run_oracle(c_oracle, scenario, cfg, p, E, steps, n_act, np.float64, SEED, ACTION_SEED)
        f32 = run_oracle(c_oracle, scenario, cfg, p, E, steps, n_act, np.float32, SEED, ACTION_SEED)
        compare(f32, f64, name)
        # study, not parity: the barrier QP as a restated cvxopt interior-point iterate at reltol = feastol = 1e-2 (what the
        # reference's rps asks of cvxopt; oracle/oracle_core.h barrier_qp_ipm) instead of the exact projection
        ipm = run_oracle(c_oracle, scenario, dict(cfg, qp_solver="cvxopt_restated"), p, E, steps, n_act, np.float64, SEED, ACTION_SEED)
        # round 5: `barrier_solver: cvxopt` is a mode of the product.  Its float tier (float32 step + ipm_spec_v0 in binary64 = the
        # HIP kernels of that mode, bit for bit) against the float64 restatement in cvxopt's own operation order: same distributions
        ipm32 = run_oracle(c_oracle, scenario, dict(cfg, barrier_solver="cvxopt"), p, E, steps, n_act, np.float32, SEED, ACTION_SEED)
        compare(ipm32, ipm, name + " (barrier_solver: cvxopt, float tier vs float64)")
        out["cases"][name] = {"envs": E, "steps": steps, "float64": f64, "float32": f32, "float64_cvxopt_restated": ipm,
                              "float32_cvxopt": ipm32}
        print(name, "collisions, exact projection vs restated cvxopt iterate:", f64["violation_counts"][1], ipm["violation_counts"][1])
        print(name, {k: f64[k] for k in ("episodes", "return_mean", "length_mean", "violation_counts")},
              {k: f32[k] for k in ("episodes", "return_mean", "length_mean", "violation_counts")})
    if "--write" in sys.argv:
        with open(os.path.join(here, "golden", "FREE_RUNNING_STATS.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
            f.write("\n")
