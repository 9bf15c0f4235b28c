"""The single-env facade (marbler_amd.wrapper.Wrapper) returns what the reference's Wrapper returns,
type for type (wrapper.py:36-44), and the gymma-shaped adapters reduce it the way EPyMARL does."""
import json
import os
import tempfile

import numpy as np
import pytest
import yaml

from helpers import GPU_NAME, GOLDEN_DIR, load_golden, pre_state

pytestmark = pytest.mark.gpu

MSG = {0: None, 1: "collision", 2: "boundary", 3: "collision_boundary"}
FILES = ["pcp_n5", "warehouse_n8", "mt_n6", "simple_n4_default", "arctic_default", "viol_PredatorCapturePrey_both",
         "viol_Warehouse_late_boundary", "viol_ArcticTransport_collision",
         # the same facade with `barrier_solver: cvxopt` in the YAML: the reference's vectors over the restated cvxopt iterate
         "ipm_pcp_n5", "ipm_warehouse_n8", "ipm_mt_n6", "ipm_arctic_default", "ipm_viol_PredatorCapturePrey_both"]


@pytest.mark.parametrize("name", FILES)
def test_wrapper_matches_reference_types_and_values(name):
    import torch
    from marbler_amd import Wrapper
    g, scenario, cfg = load_golden(os.path.join(GOLDEN_DIR, name + ".npz"))
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(cfg, f)
        path = f.name
    try:
        w = Wrapper(scenario, path)
    finally:
        os.unlink(path)
    N = w.n_agents
    assert len(w.action_space) == N and len(w.observation_space) == N
    obs0 = w.reset()
    assert isinstance(obs0, list) and len(obs0) == N and all(v == 0 for v in obs0[0])
    st = pre_state(g)
    rows = [t for t in range(len(g["actions"]))][:60]
    rows += [int(t) for t in np.nonzero(g["done"])[0][:4]]
    import parity
    tol = parity.TOL                                                   # 1e-5 in every scenario
    for t in rows:
        w.env.vec.load_state_dict({GPU_NAME.get(k, k): torch.as_tensor(np.asarray(v[t:t + 1])) for k, v in st.items()})
        obs, rew, done, info = w.step([int(a) for a in g["actions"][t]])
        assert isinstance(obs, tuple) and len(obs) == N and obs[0].dtype == np.float64
        assert isinstance(rew, list) and isinstance(rew[0], float) and isinstance(done, list) and isinstance(done[0], bool)
        assert all(d == bool(g["done"][t]) for d in done)
        assert np.abs(np.array(rew) - g["reward"][t]).max() <= 1e-5
        assert np.abs(info["dist_travelled"] - g["dist"][t]).max() <= tol and info["dist_travelled"].dtype == np.float64
        assert info.get("message") == MSG[int(g["viol"][t])]
        assert info.get("remaining", -1) == int(g["remaining"][t])
        got_obs = np.array(obs)
        for a in np.nonzero(np.abs(got_obs - g["obs"][t]).max(axis=1) > tol)[0]:   # only a float64 near-tie may differ
            why, _ = parity.explain_row(scenario, cfg, int(a), got_obs, g["obs"][t], g["post_poses"][t],
                                        g["post_prey_loc"][t] if "post_prey_loc" in g.files else None,
                                        g["post_prey_captured"][t] if "post_prey_captured" in g.files else None)
            assert why is None, (t, int(a), why)
        assert w.env.agent_poses.shape == (3, N)
    w.close()


def test_gymma_vec_env_contract():
    import torch
    from marbler_amd.gymma import GymmaVecEnv
    E = 64
    env = GymmaVecEnv("robotarium_gym:PredatorCapturePrey-v0", E, time_limit=30,
                      overrides={"predator": 3, "capture": 2, "n_agents": 5})
    info = env.get_env_info()
    assert info == {"state_shape": 80, "obs_shape": 16, "n_actions": 5, "n_agents": 5, "episode_limit": 30}
    obs, state = env.reset()
    assert obs.shape == (E, 5, 16) and state.shape == (E, 80) and float(obs.abs().max()) == 0
    assert env.get_avail_actions().shape == (E, 5, 5) and int(env.get_avail_actions().min()) == 1
    g = torch.Generator(device=obs.device)
    g.manual_seed(0)
    ended_total = 0
    el = torch.zeros(E, dtype=torch.int64, device=obs.device)
    for t in range(70):
        a = torch.randint(0, 5, (E, 5), generator=g, device=obs.device, dtype=torch.int32)
        r, term, inf = env.step(a)
        assert r.shape == (E,) and term.shape == (E,) and term.dtype == torch.bool
        assert torch.allclose(r, env.env.reward.sum(1))                     # float(sum(reward_n))
        el += 1
        scen_done = env.env.done_u8.bool()
        assert torch.equal(term, scen_done | (el >= 30))                    # gym TimeLimit on top of the scenario
        assert torch.equal(inf["TimeLimit.truncated"], (el >= 30) & ~scen_done)
        el[term] = 0
        ended_total += int(term.sum())
        # an ended env shows the reset observation (zeros) next; a running one its own position
        o = env.get_obs()
        assert float(o[term].abs().max() if term.any() else 0) == 0
        run = ~term
        assert torch.equal(o[run][:, :, 0], env.env.poses[run][:, 0])
        assert torch.equal(env.get_state(), o.reshape(E, -1))
    assert ended_total >= 2 * E
    env.close()


def test_gymma_single_env_types():
    from marbler_amd.gymma import GymmaEnv
    env = GymmaEnv("robotarium_gym:Warehouse-v0", time_limit=5)
    obs, state = env.reset()
    assert len(obs) == env.n_agents and state.shape == (env.get_state_size(),)
    for t in range(5):
        r, done, info = env.step([1] * env.n_agents)
        assert isinstance(r, float) and isinstance(done, bool)
    assert done and info.get("TimeLimit.truncated")
    assert env.get_env_info()["n_actions"] == 5 and len(env.get_avail_actions()) == env.n_agents
    env.close()


def test_batched_runner_collects_transitions():
    """The time-major transition tensors of the batched runner: shapes, the gymma reductions, and
    consistency with the env's own episode statistics."""
    import torch
    from marbler_amd.evaluate import BatchedActor
    from marbler_amd.gymma import BatchedRunner, GymmaVecEnv
    from test_gpu_actor import _random_actor
    E, T = 128, 120
    v = GymmaVecEnv("robotarium_gym:PredatorCapturePrey-v0", E, time_limit=1000, seed=3)
    actor = BatchedActor(_random_actor(1, v.obs_size + v.n_agents, 64, v.n_actions, True, seed=2), v.n_agents,
                         device=v.env.device)
    runner = BatchedRunner(v, actor, epsilon=0.3, seed=1)
    b = runner.run(T)
    N, D, A = v.n_agents, v.obs_size, v.n_actions
    assert tuple(b["obs"].shape) == (T + 1, E, N, D) and tuple(b["state"].shape) == (T + 1, E, N * D)
    assert tuple(b["actions"].shape) == (T, E, N) and tuple(b["reward"].shape) == (T, E)
    assert tuple(b["avail_actions"].shape) == (T + 1, E, N, A) and bool((b["avail_actions"] == 1).all())
    assert int(b["actions"].min()) >= 0 and int(b["actions"].max()) < A
    assert torch.equal(b["state"], b["obs"].reshape(T + 1, E, N * D))
    # an episode starts where the previous step ended one (and at t = 0); its first observation is zeros
    assert bool(b["episode_start"][0].all())
    assert torch.equal(b["episode_start"][1:], b["terminated"][:-1])
    assert bool((b["obs"][:-1][b["episode_start"]] == 0).all())
    # the rewards of finished episodes add up to the env's own accumulators (shared reward: N x reward[0])
    n_done = int(b["terminated"].sum())
    assert n_done == int(v.env.done_count.sum()) and n_done > E
    ret_env = float(v.env.done_return_sum.sum()) + float(v.env.ep_return.sum())
    assert abs(float(b["reward"].sum()) / N - ret_env) < 1e-2 * max(1.0, abs(ret_env))
    # a second call continues the same episodes
    b2 = runner.run(10)
    assert torch.equal(b2["episode_start"][0], b["terminated"][-1])


@pytest.mark.parametrize("kernel", ["group", "tpe"])
@pytest.mark.parametrize("key,ov,limit,epsilon", [
    ("robotarium_gym:PredatorCapturePrey-v0", {}, 25, 0.0),
    ("robotarium_gym:PredatorCapturePrey-v0", {}, 25, 0.25),
    ("robotarium_gym:Warehouse-v0", {}, 30, 0.25),                 # rows of 18 floats: not whole 16-byte units
    ("robotarium_gym:MaterialTransport-v0", {}, 12, 0.25),         # 9-float rows, 20 actions
    ("robotarium_gym:PredatorCapturePrey-v0", {"predator": 3, "capture": 2, "n_agents": 5}, 25, 0.1),
    ("robotarium_gym:PredatorCapturePrey-v0", {"barrier_solver": "cvxopt"}, 25, 0.25),       # the gymma block of the interior-point kernels
    ("robotarium_gym:Warehouse-v0", {"barrier_solver": "cvxopt", "n_agents": 6}, 30, 0.1)])
def test_batched_runner_in_place_path_equals_the_composed_one(key, ov, limit, epsilon, kernel, monkeypatch):
    """BatchedRunner over a fused GymmaVecEnv is TWO launches per time step: the env step writes the next observation (zeros for
    an env that ended: rg_step_io.zero_obs_on_end), the summed reward and the episode-end flags straight into the transition
    batch, the actor reads them there.  Over a composed one (fused=False: gym's TimeLimit and the reductions as torch ops around
    the step) it goes through step() / get_obs().  Same envs, same actor, same exploration stream: the batches must be equal,
    element for element (a short time limit, so that truncations happen), on both step kernels."""
    import torch
    from marbler_amd.evaluate import BatchedActor
    from marbler_amd.gymma import BatchedRunner, GymmaVecEnv
    from test_gpu_actor import _random_actor
    monkeypatch.setenv("RG_STEP_KERNEL", kernel)
    E, T = 96, 70
    batches = []
    for fused in (True, False):
        v = GymmaVecEnv(key, E, time_limit=limit, seed=5, fused=fused, overrides=ov or None)
        actor = BatchedActor(_random_actor(1, v.obs_size + v.n_agents, 64, v.n_actions, True, seed=4), v.n_agents, device=v.env.device)
        runner = BatchedRunner(v, actor, epsilon=epsilon, seed=9)
        b = runner.run(T)
        b2 = runner.run(7)       # a second call continues the same episodes
        obs_after = v.get_obs().clone()
        batches.append((b, b2, int(v.env.done_count.sum()), obs_after))
        v.env.close()
    (a, a2, na, oa), (c, c2, nc, oc) = batches
    assert na == nc and int(a["terminated"].sum()) > E // 2
    assert torch.equal(oa, oc)
    for x, y in ((a, c), (a2, c2)):
        for k in ("obs", "state", "actions", "terminated", "episode_start", "avail_actions"):
            assert torch.equal(x[k], y[k]), k
        # the summed reward: the launch adds the agents' rewards in agent order, torch's reduction of the composed path in its own
        # order -- equal to the last bit for 4 agents, within an ulp of the sum for 5 and more
        assert torch.allclose(x["reward"], y["reward"], rtol=0, atol=2e-6), "reward"


@pytest.mark.parametrize("name", ["pcp_n5", "warehouse_n8", "mt_n6", "viol_PredatorCapturePrey_collision", "ipm_pcp_n5", "ipm_warehouse_n8"])
def test_gymma_env_reduces_the_reference_vectors_like_epymarl(name):
    """GymmaEnv (EPyMARL's gymma contract: float(sum(reward_n)), all(done_n), padded per-agent observations,
    state = their concatenation) fed the reference's own states and actions must return the reductions of the
    reference's own outputs (tests/golden, captured from the reference Wrapper)."""
    import torch
    import parity
    from marbler_amd.gymma import GymmaEnv
    g, scenario, cfg = load_golden(os.path.join(GOLDEN_DIR, name + ".npz"))
    env = GymmaEnv(f"robotarium_gym:{scenario}-v0", time_limit=10 ** 6, seed=1, **{k: v for k, v in cfg.items() if k not in ("seed", "device")})   # the YAML's `device` is the torch device of the reference's actor
    env.reset()
    st = pre_state(g)
    N, D = env.n_agents, env.get_obs_size()
    assert env.get_env_info()["state_shape"] == N * D and g["obs"].shape[1:] == (N, D)
    for t in list(range(min(40, len(g["actions"])))) + [int(i) for i in np.nonzero(g["done"])[0][:3]]:
        env._v.env.load_state_dict({GPU_NAME.get(k, k): torch.as_tensor(np.asarray(v[t:t + 1])) for k, v in st.items()})
        r, done, info = env.step([int(a) for a in g["actions"][t]])
        assert isinstance(r, float) and abs(r - float(np.sum(g["reward"][t]))) <= 1e-4
        assert done == bool(g["done"][t])
        obs = np.stack(env.get_obs())
        for a in np.nonzero(np.abs(obs - g["obs"][t]).max(axis=1) > parity.TOL)[0]:
            why, _ = parity.explain_row(scenario, cfg, int(a), obs.astype(np.float64), g["obs"][t], g["post_poses"][t],
                                        g["post_prey_loc"][t] if "post_prey_loc" in g.files else None,
                                        g["post_prey_captured"][t] if "post_prey_captured" in g.files else None)
            assert why is None, (t, int(a), why)
        assert np.array_equal(env.get_state(), obs.reshape(-1))
        assert info.get("message") == MSG[int(g["viol"][t])]
    env.close()


@pytest.mark.parametrize("kernel", ["group", "tpe"])
@pytest.mark.parametrize("key,ov,n_act,limit", [
    ("robotarium_gym:PredatorCapturePrey-v0", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 9),
    ("robotarium_gym:Warehouse-v0", {"n_agents": 8}, 5, 1000),
    ("robotarium_gym:MaterialTransport-v0", {}, 20, 7)])
def test_fused_time_limit_equals_the_composed_gymma_step(kernel, key, ov, n_act, limit, monkeypatch):
    """gym's TimeLimit and the gymma reductions inside the step launch (rg_step_io's gymma block) against the same
    contract composed from torch ops around the plain step: rewards, terminated / truncated flags, the observations a
    consumer sees next, the episode statistics and the env state, step after step through truncations and resets."""
    import torch
    from marbler_amd.gymma import GymmaVecEnv
    monkeypatch.setenv("RG_STEP_KERNEL", kernel)
    E = 256
    a = GymmaVecEnv(key, E, time_limit=limit, overrides=ov, seed=5, fused=True)
    b = GymmaVecEnv(key, E, time_limit=limit, overrides=ov, seed=5, fused=False)
    assert a.env.time_limit == limit and b.env.time_limit == 0
    a.reset()
    b.reset()
    g = torch.Generator(device=a.env.device)
    g.manual_seed(3)
    n_trunc = n_done = 0
    for t in range(90):
        act = torch.randint(0, n_act, (E, a.n_agents), generator=g, device=a.env.device, dtype=torch.int32)
        ra, ta, ia = a.step(act)
        rb, tb, ib = b.step(act)
        assert torch.equal(ta, tb) and torch.equal(ia["TimeLimit.truncated"], ib["TimeLimit.truncated"]), t
        assert torch.allclose(ra, rb, rtol=0, atol=1e-5), t
        assert torch.equal(a.get_obs().view(torch.int32), b.get_obs().view(torch.int32)), t
        assert torch.equal(a.get_state(), a.get_obs().reshape(E, -1))
        for k in ("violation", "remaining", "dist_travelled"):
            assert torch.equal(ia[k], ib[k]), (t, k)
        assert torch.equal(a.env.poses.view(torch.int32), b.env.poses.view(torch.int32)), t
        assert torch.equal(a.env.episode_steps, b.env.episode_steps) and torch.equal(a.env.reset_count, b.env.reset_count)
        n_trunc += int(ia["TimeLimit.truncated"].sum())
        n_done += int(ta.sum())
    sa, sb = a.get_stats(), b.get_stats()
    assert sa["episodes"] == sb["episodes"] == n_done and sa["steps"] == sb["steps"]
    assert abs(sa["return_sum"] - sb["return_sum"]) < 1e-3 * max(1.0, abs(sb["return_sum"]))
    assert n_done > E // 2 and (n_trunc > E if limit < 50 else n_trunc == 0)
    a.close()
    b.close()


def test_gymma_step_outputs_survive_the_next_step():
    """A runner appends step t's reward / terminated tensors to a list while it takes step t + 1 (EPyMARL's episode
    batch): the fused step must hand out fresh tensors, not views of the buffers the next launch overwrites.  With
    alias_outputs=True the views themselves are returned (documented), and the list then repeats the last step."""
    import torch
    from marbler_amd.gymma import GymmaVecEnv
    key, ov, E, T = "robotarium_gym:PredatorCapturePrey-v0", {"predator": 3, "capture": 2, "n_agents": 5}, 128, 40
    g = torch.Generator(device="cuda:0")
    g.manual_seed(9)
    acts = torch.randint(0, 5, (T, E, 5), generator=g, device="cuda:0", dtype=torch.int32)
    kept = {}
    for alias in (False, True):
        env = GymmaVecEnv(key, E, time_limit=12, overrides=ov, seed=2, fused=True, alias_outputs=alias)
        env.reset()
        rows = [env.step(acts[t]) for t in range(T)]
        kept[alias] = (torch.stack([r[0] for r in rows]), torch.stack([r[1] for r in rows]),
                       torch.stack([r[2]["TimeLimit.truncated"] for r in rows]))
        env.close()
    ref = GymmaVecEnv(key, E, time_limit=12, overrides=ov, seed=2, fused=False)   # composed contract: always fresh tensors
    ref.reset()
    rows = [ref.step(acts[t]) for t in range(T)]
    want = (torch.stack([r[0] for r in rows]), torch.stack([r[1] for r in rows]),
            torch.stack([r[2]["TimeLimit.truncated"] for r in rows]))
    ref.close()
    assert torch.allclose(kept[False][0], want[0], rtol=0, atol=1e-5)
    assert torch.equal(kept[False][1], want[1]) and torch.equal(kept[False][2], want[2])
    assert int(want[1].sum()) > 0 and int(want[2].sum()) > 0
    # the aliasing form: every kept entry is the same buffer = the last step
    assert torch.equal(kept[True][1], want[1][-1:].expand_as(want[1]))


def test_real_time_paces_the_single_env_wrapper():
    """`real_time: True` (rps sim_in_real_time: 0.033 s of wall clock per simulator iteration): the facade returns a step no sooner than
    update_frequency x 0.033 s after the call, with the same values as the unpaced env; the batched engine still refuses the key."""
    import time
    from marbler_amd import VecRobotariumEnv, Wrapper
    from marbler_amd.params import default_config_path, load_config
    cfg = load_config("Warehouse", default_config_path("Warehouse"), {"real_time": True, "update_frequency": 6, "seed": 3})
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(cfg, f)
        path = f.name
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(dict(cfg, real_time=False), f)
        path_fast = f.name
    try:
        slow, fast = Wrapper("Warehouse", path), Wrapper("Warehouse", path_fast)
        assert slow.env.args.real_time is True
        with pytest.raises(ValueError):
            VecRobotariumEnv("Warehouse", 4, config_path=path)
    finally:
        os.unlink(path)
        os.unlink(path_fast)
    slow.reset()
    fast.reset()
    acts = [1, 0, 3, 2, 1, 0]
    slow.step(acts)                                   # (first call: lazily created host buffers)
    fast.step(acts)
    t0 = time.monotonic()
    o1, r1, d1, i1 = slow.step(acts)
    dt_slow = time.monotonic() - t0
    t0 = time.monotonic()
    o2, r2, d2, i2 = fast.step(acts)
    dt_fast = time.monotonic() - t0
    assert dt_slow >= 6 * 0.033 - 1e-3 and dt_slow < 6 * 0.033 + 0.05 and dt_fast < 0.05, (dt_slow, dt_fast)
    assert all(np.array_equal(a, b) for a, b in zip(o1, o2)) and r1 == r2 and d1 == d2
    slow.close()
    fast.close()
