"""Free-running rollouts on the GPU (auto-reset on) against the float32 oracle stepping the same
envs on the CPU with the oracle's reset twin: every output of every step, bit for bit, over whole
episodes including violations, captures, loading/unloading and resets.  Plus size-independent
properties at the bench size (4096 envs)."""
import numpy as np
import pytest

from helpers import oracle_reset as _oracle_reset, oracle_reset_params

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("step_kernel")]

CASES = [("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 260),
         ("PredatorCapturePrey", {"num_neighbors": 2, "capability_aware": True}, 5, 120),
         ("Warehouse", {"n_agents": 8}, 5, 230),
         ("Warehouse", {}, 5, 120),
         ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20, 160),
         ("MaterialTransport", {}, 20, 100),
         ("PredatorCapturePrey", {"predator": 6, "capture": 6, "n_agents": 12, "num_prey": 10, "start_dist": 0.25,
                                  "num_neighbors": 4}, 5, 60),
         ("PredatorCapturePrey", {"predator": 3, "capture": 2, "collision_variant": "center"}, 5, 150),
         ("Warehouse", {"n_agents": 8, "barrier_certificate": "default"}, 5, 150),
         ("PredatorCapturePrey", {"predator": 3, "capture": 2, "penalize_violations": False}, 5, 120),
         ("MaterialTransport", {"capability_aware": True, "qp_max_sweeps": 6}, 20, 80),
         ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "num_prey": 12}, 5, 150),   # > 8 prey: LDS / loop paths
         ("PredatorCapturePrey", {"predator": 2, "capture": 2, "n_agents": 4, "num_prey": 33, "step_dist": 0.16}, 5, 100),  # > 32 prey: both flag words
         ("Simple", {}, 5, 130),
         ("Simple", {"n_agents": 6}, 5, 80),
         ("ArcticTransport", {}, 5, 200)]


# ragged and extreme shapes: batch sizes that do not fill a wavefront (1, 7, 65, 130 envs), the
# smallest and largest agent counts (2, 16), the largest prey count a reset grid of <= 64 cells admits here (54),
# no neighbours at all, one prey
EDGE_CASES = [("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 150, 1),
              ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 150, 7),
              ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 150, 65),
              ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 150, 130),
              ("PredatorCapturePrey", {"predator": 1, "capture": 1, "n_agents": 2, "num_neighbors": 0}, 5, 200, 65),
              ("PredatorCapturePrey", {"predator": 2, "capture": 1, "n_agents": 3, "num_prey": 1, "num_neighbors": 1}, 5, 150, 33),
              ("PredatorCapturePrey", {"predator": 8, "capture": 8, "n_agents": 16, "num_prey": 54, "start_dist": 0.2,
                                       "step_dist": 0.16, "num_neighbors": 15}, 5, 40, 9),
              ("Warehouse", {"n_agents": 2, "num_neighbors": 1}, 5, 200, 65),
              ("Warehouse", {"n_agents": 7}, 5, 150, 65),
              # more neighbour slots than other agents: rows wider than the agents fill (the tail stays zero)
              ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "num_neighbors": 5}, 5, 100, 65),
              ("PredatorCapturePrey", {"predator": 4, "capture": 4, "n_agents": 8, "num_neighbors": 8}, 5, 80, 65),
              ("Warehouse", {"n_agents": 3}, 5, 150, 65),
              ("MaterialTransport", {}, 20, 100, 65),
              ("Simple", {"n_agents": 2}, 5, 130, 65),
              ("ArcticTransport", {}, 5, 200, 65)]


@pytest.mark.parametrize("scenario,ov,n_act,steps,E", EDGE_CASES)
def test_rollout_bit_exact_ragged_and_extreme_shapes(scenario, ov, n_act, steps, E, oracle_lib):
    _rollout_bit_exact(scenario, ov, n_act, steps, oracle_lib, E)


@pytest.mark.parametrize("scenario,ov,n_act,steps", CASES)
def test_rollout_bit_exact(scenario, ov, n_act, steps, oracle_lib):
    _rollout_bit_exact(scenario, ov, n_act, steps, oracle_lib, 192)


def _rollout_bit_exact(scenario, ov, n_act, steps, oracle_lib, E, require_done=True):
    import torch
    from marbler_amd import VecRobotariumEnv
    seed = 99
    env = VecRobotariumEnv(scenario, E, overrides=ov, seed=seed, auto_reset=True, collect_qp_stats=True)
    cfg = dict(env.cfg)
    orc = oracle_lib.OracleVecEnv(scenario, cfg, E, dtype=np.float32)
    rp = oracle_reset_params(oracle_lib, env.params)
    env.reset()
    episodes = np.zeros(E, np.int64)
    for e in range(E):
        _oracle_reset(oracle_lib, orc, rp, seed, e, 0)
    rng = np.random.RandomState(5)
    n_done = n_viol = 0
    ret = np.zeros(E, np.float32)
    ret_sum = np.zeros(E, np.float32)
    for t in range(steps):
        a = rng.randint(0, n_act, size=(E, env.N)).astype(np.int32)
        obs, rew, done, info = env.step(torch.as_tensor(a, device=env.device))
        o_obs, o_rew, o_done, o_info = orc.step(a)
        g = {"obs": obs.cpu().numpy(), "reward": rew.cpu().numpy(), "done": done.cpu().numpy().astype(np.uint8),
             "dist": info["dist_travelled"].cpu().numpy(), "viol": info["violation"].cpu().numpy(),
             "remaining": info["remaining"].cpu().numpy()}
        assert np.array_equal(g["done"], o_done), t
        assert np.array_equal(g["viol"], o_info["violation"]), t
        assert np.array_equal(g["remaining"], o_info["remaining"]), t
        assert np.array_equal(g["obs"].view(np.uint32), o_obs.view(np.uint32)), t
        assert np.array_equal(g["reward"].view(np.uint32), o_rew.view(np.uint32)), t
        assert np.array_equal(g["dist"].view(np.uint32), o_info["dist_travelled"].view(np.uint32)), t
        assert np.array_equal(env.qp_sweeps.cpu().numpy(), orc.qp_sweeps), t
        # rollout statistics (misc.py:178-185)
        r = o_rew[:, 0] if env.params.shared_reward else o_rew.sum(axis=1, dtype=np.float32) if False else None
        if env.params.shared_reward:
            ret = ret + o_rew[:, 0]
        else:
            s = np.zeros(E, np.float32)
            for k in range(env.N):
                s = s + o_rew[:, k]
            ret = ret + s
        for e in np.nonzero(o_done)[0]:
            ret_sum[e] = ret_sum[e] + ret[e]
            ret[e] = 0
            episodes[e] += 1
            _oracle_reset(oracle_lib, orc, rp, seed, e, int(episodes[e]))
        n_done += int(o_done.sum())
        n_viol += int((o_info["violation"] > 0).sum())
        # state after the (possibly reset) step
        assert np.array_equal(env.poses.cpu().numpy().view(np.uint32), orc.poses.view(np.uint32)), t
        assert np.array_equal(env.episode_steps.cpu().numpy(), orc.steps), t
    assert n_done > 0 or E < 8 or not require_done
    assert np.array_equal(env.done_count.cpu().numpy(), episodes)
    assert np.array_equal(env.done_return_sum.cpu().numpy().view(np.uint32), ret_sum.view(np.uint32))
    assert np.array_equal(env.ep_return.cpu().numpy().view(np.uint32), ret.view(np.uint32))
    env.close()


def test_bench_size_properties():
    """4096 x 5 (BASELINE configs[1]): properties that need no oracle."""
    import torch
    from marbler_amd import VecRobotariumEnv
    E = 4096
    env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=3)
    env.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(0)
    total_done = 0
    for t in range(120):
        a = torch.randint(0, 5, (E, 5), generator=g, device=env.device, dtype=torch.int32)
        prev = env.poses.clone()
        steps_before = env.episode_steps.clone()
        obs, rew, done, info = env.step(a)
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        # a robot moves at most 29 * 0.033 * 0.2 m per step; dist_travelled obeys the same bound
        live = ~done
        d = (env.poses[live][:, :2] - prev[live][:, :2]).norm(dim=1)
        assert float(d.max()) <= 29 * 0.033 * 0.2 + 1e-5
        assert float(info["dist_travelled"].max()) <= 29 * 0.033 * 0.2 + 0.0067 + 1e-5
        assert float(env.poses[:, 2].abs().max()) <= np.pi + 1e-6
        # shared reward: identical across agents; violation <=> reward -5 and done
        assert torch.equal(rew, rew[:, :1].expand_as(rew))
        v = info["violation"] > 0
        assert bool((rew[v, 0] == -5).all()) and bool(done[v].all())
        # own block of the observation is the agent's position (of the terminal state for done envs)
        assert torch.equal(obs[live][:, :, 0], env.poses[live][:, 0]) and torch.equal(obs[live][:, :, 1], env.poses[live][:, 1])
        # step counters: +1, or 0 after an auto-reset
        assert torch.equal(env.episode_steps[live], steps_before[live] + 1)
        assert int(env.episode_steps[done].abs().sum()) == 0
        total_done += int(done.sum())
    assert total_done > 0
    assert int(env.done_count.sum()) == total_done
    env.close()


def test_step_is_deterministic_and_shard_invariant():
    import torch
    from marbler_amd import VecRobotariumEnv
    ov = {"n_agents": 8}
    a = torch.randint(0, 5, (60, 512, 8), device="cuda:0", dtype=torch.int32)
    full = VecRobotariumEnv("Warehouse", 512, overrides=ov, seed=11)
    lo = VecRobotariumEnv("Warehouse", 256, overrides=ov, seed=11, env_offset=0)
    hi = VecRobotariumEnv("Warehouse", 256, overrides=ov, seed=11, env_offset=256)
    for env in (full, lo, hi):
        env.reset()
    for t in range(60):
        o, r, d, _ = full.step(a[t])
        o1, r1, d1, _ = lo.step(a[t, :256].contiguous())
        o2, r2, d2, _ = hi.step(a[t, 256:].contiguous())
        assert torch.equal(o, torch.cat([o1, o2])) and torch.equal(r, torch.cat([r1, r2])) and \
            torch.equal(d, torch.cat([d1, d2]))
    assert torch.equal(full.done_return_sum, torch.cat([lo.done_return_sum, hi.done_return_sum]))


@pytest.mark.parametrize("scenario,ov,n_act,E,auto_kernel", [
    ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 65536, "tpe"),    # the threshold itself
    ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 61440, "group"),  # just below it
    ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20, 98304, "tpe"),
    ("Warehouse", {"n_agents": 8}, 5, 65536, "group")])                                        # N >= 7: always the lane-group kernel
def test_large_batch_dispatch_matches_lane_group_kernel(scenario, ov, n_act, E, auto_kernel, monkeypatch):
    """The library's own kernel choice at large batches (robogym_capi.hip: tpe_min_envs; `rg_step_kernel` says which one
    a handle launches) and the thread-per-env kernel forced, against the lane-group kernel forced: every output and the
    whole state bit for bit, over steps that include auto-resets.  Warehouse 65536 x 8: N >= 7 runs the lane-group kernel
    whatever is asked for (the thread-per-env instantiations for N = 7, 8 left the library in round 4)."""
    import torch
    from marbler_amd import VecRobotariumEnv
    monkeypatch.delenv("RG_STEP_KERNEL", raising=False)
    auto = VecRobotariumEnv(scenario, E, overrides=ov, seed=5)
    assert auto.step_kernel == auto_kernel
    monkeypatch.setenv("RG_STEP_KERNEL", "tpe")
    tpe = VecRobotariumEnv(scenario, E, overrides=ov, seed=5)
    assert tpe.step_kernel == ("tpe" if auto.N <= 6 else "group")   # N >= 7 has no thread-per-env instantiation (round 4)
    monkeypatch.setenv("RG_STEP_KERNEL", "group")
    ref = VecRobotariumEnv(scenario, E, overrides=ov, seed=5)
    assert ref.step_kernel == "group"
    g = torch.Generator(device=auto.device)
    g.manual_seed(2)
    envs = (auto, tpe, ref)
    for env in envs:
        env.reset()
    for t in range(40):
        a = torch.randint(0, n_act, (E, auto.N), generator=g, device=auto.device, dtype=torch.int32)
        outs = [env.step(a) for env in envs]
        o2, r2, d2, i2 = outs[2]
        for o1, r1, d1, i1 in outs[:2]:
            assert torch.equal(o1.view(torch.int32), o2.view(torch.int32)), t
            assert torch.equal(r1.view(torch.int32), r2.view(torch.int32)) and torch.equal(d1, d2), t
            for k in i1:
                assert torch.equal(i1[k], i2[k]), (t, k)
    s2 = ref.state_dict()
    for env in (auto, tpe):
        s1 = env.state_dict()
        for k in s1:
            assert torch.equal(s1[k], s2[k]), k
    assert int(auto.done_count.sum()) > 0
    for env in envs:
        env.close()


@pytest.mark.parametrize("scenario,ov,n_act,E", [
    ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 1000),
    ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 66000),   # thread-per-env: past its multi-step kernel's range
    ("Warehouse", {"n_agents": 8}, 5, 300),
    ("MaterialTransport", {}, 20, 129),
    ("PredatorCapturePrey", {"predator": 6, "capture": 6, "n_agents": 12, "num_prey": 10, "start_dist": 0.25,
                             "num_neighbors": 4}, 5, 40),
    ("ArcticTransport", {}, 5, 200)])
def test_rollout_entry_equals_repeated_steps(scenario, ov, n_act, E):
    """rg_rollout (K steps in one launch, envs advancing independently) against K rg_step launches:
    every per-step output and the final state, bit for bit, across auto-resets."""
    _rollout_equals_steps(scenario, ov, n_act, E)


def _rollout_equals_steps(scenario, ov, n_act, E, K=48, reps=3, require_done=True, time_limit=None):
    import torch
    from marbler_amd import VecRobotariumEnv
    a = VecRobotariumEnv(scenario, E, overrides=ov, seed=21)
    b = VecRobotariumEnv(scenario, E, overrides=ov, seed=21)
    g = torch.Generator(device=a.device)
    g.manual_seed(4)
    a.reset()
    b.reset()
    for rep in range(reps):   # several launches of K steps: the state carries over between launches
        acts = torch.randint(0, n_act, (K, E, a.N), generator=g, device=a.device, dtype=torch.int32)
        out = a.rollout(acts)
        for k in range(K):
            obs, rew, done, info = b.step(acts[k])
            assert torch.equal(out["obs"][k].view(torch.int32), obs.view(torch.int32)), (rep, k)
            assert torch.equal(out["reward"][k].view(torch.int32), rew.view(torch.int32)), (rep, k)
            assert torch.equal(out["done"][k].bool(), done), (rep, k)
            assert torch.equal(out["dist_travelled"][k].view(torch.int32), info["dist_travelled"].view(torch.int32))
            assert torch.equal(out["violation"][k], info["violation"]) and torch.equal(out["remaining"][k], info["remaining"])
    sa, sb = a.state_dict(), b.state_dict()
    for key in sa:
        assert torch.equal(sa[key], sb[key]), key
    assert torch.equal(a.done_count, b.done_count) and (int(a.done_count.sum()) > 0 or not require_done)
    assert torch.equal(a.done_return_sum.view(torch.int32), b.done_return_sum.view(torch.int32))
    a.close()
    b.close()
