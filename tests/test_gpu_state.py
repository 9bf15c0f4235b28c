"""Env-state snapshots, episode bookkeeping across explicit resets, seeding and device / stream handling
(the -m gpu tier; rows a17, f4 and the boundary's device contract)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, load_golden

pytestmark = pytest.mark.gpu

SNAP_CASES = [("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5),
              ("Warehouse", {"n_agents": 8}, 5),
              ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20),
              ("ArcticTransport", {}, 5)]


@pytest.mark.parametrize("scenario,ov,n_act", SNAP_CASES)
def test_snapshot_restores_and_continues_bit_identically(scenario, ov, n_act, step_kernel):
    """step 40, snapshot, step 40 -> A; restore (into a FRESH env object), step 40 -> B; A == B bit for bit:
    every output of every step, the final state, auto-resets and the episode statistics."""
    import torch
    from marbler_amd import VecRobotariumEnv
    E = 320
    a = VecRobotariumEnv(scenario, E, overrides=ov, seed=17)
    g = torch.Generator(device=a.device)
    g.manual_seed(3)
    acts = torch.randint(0, n_act, (80, E, a.N), generator=g, device=a.device, dtype=torch.int32)
    a.reset()
    for t in range(40):
        a.step(acts[t])
    snap = {k: v.cpu() for k, v in a.state_dict().items()}          # a host copy, as torch.save would hold
    assert int(a.done_count.sum()) > 0 and float(a.ep_return.abs().sum()) > 0   # mid-episode, statistics running
    outs_a = []
    for t in range(40, 80):
        o, r, d, i = a.step(acts[t])
        outs_a.append([x.clone() for x in (o, r, d, i["dist_travelled"], i["violation"], i["remaining"])])
    b = VecRobotariumEnv(scenario, E, overrides=ov, seed=999)       # a different key: the snapshot carries its own
    b.load_state_dict(snap)
    for t in range(40, 80):
        o, r, d, i = b.step(acts[t])
        for x, y in zip(outs_a[t - 40], (o, r, d, i["dist_travelled"], i["violation"], i["remaining"])):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8)), t
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa) == set(sb) and {"ep_return", "done_return_sum", "done_count", "done_steps_sum", "reset_count"} <= set(sa)
    for k in sa:
        assert torch.equal(sa[k].view(torch.uint8), sb[k].view(torch.uint8)), k
    assert int(b.done_count.sum()) > int(snap["done_count"].sum())  # episodes ended, and were reset, after the restore


def test_snapshot_of_a_time_limited_env_keeps_the_time_limit_counter(step_kernel):
    import torch
    from marbler_amd import VecRobotariumEnv
    ov = {"predator": 3, "capture": 2, "n_agents": 5}
    a = VecRobotariumEnv("PredatorCapturePrey", 128, overrides=ov, seed=3)
    a.enable_time_limit(11)
    acts = torch.randint(0, 5, (60, 128, 5), device=a.device, dtype=torch.int32)
    a.reset()
    for t in range(17):
        a.step(acts[t])
    snap = {k: v.cpu() for k, v in a.state_dict().items()}
    assert "elapsed" in snap and int(snap["elapsed"].max()) > 0
    ends_a = []
    for t in range(17, 60):
        a.step(acts[t])
        ends_a.append((a.ended.clone(), a.truncated.clone(), a.reward_sum.clone()))
    b = VecRobotariumEnv("PredatorCapturePrey", 128, overrides=ov, seed=77)
    b.enable_time_limit(11)
    b.load_state_dict(snap)
    for t in range(17, 60):
        b.step(acts[t])
        e, tr, rs = ends_a[t - 17]
        assert torch.equal(b.ended, e) and torch.equal(b.truncated, tr) and torch.equal(b.reward_sum.view(torch.int32), rs.view(torch.int32)), t
    assert torch.equal(a.elapsed, b.elapsed) and torch.equal(a.poses.view(torch.int32), b.poses.view(torch.int32))
    assert int(sum(int(x[1].sum()) for x in ends_a)) > 0       # the limit did fire
    # the counter is part of the contract in both directions: no silent drop, no stale counter
    c = VecRobotariumEnv("PredatorCapturePrey", 128, overrides=ov, seed=5)          # no time limit
    with pytest.raises(KeyError, match="elapsed"):
        c.load_state_dict(snap)
    c.load_state_dict({k: v for k, v in snap.items() if k != "elapsed"})            # dropping it explicitly is fine
    plain = {k: v.cpu() for k, v in c.state_dict().items()}
    assert "elapsed" not in plain
    with pytest.raises(KeyError, match="elapsed"):
        b.load_state_dict(plain)                                                    # time-limited env needs the counter
    b.load_state_dict(dict(plain, elapsed=torch.zeros(128, dtype=torch.int32)))
    assert int(b.elapsed.abs().sum()) == 0
    with pytest.raises(KeyError, match="unknown"):
        c.load_state_dict({"no_such_array": torch.zeros(1)})


def test_explicit_reset_restarts_the_running_return_and_can_book_the_episode(step_kernel):
    import torch
    from marbler_amd import VecRobotariumEnv
    E = 96
    env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=4,
                           auto_reset=True)
    env.reset()
    a = torch.full((E, 5), 4, dtype=torch.int32, device=env.device)         # nobody moves: no violations, no captures
    for _ in range(3):
        env.step(a)
    assert int(env.done_count.sum()) == 0
    ret = env.ep_return.clone()
    assert torch.allclose(ret, torch.full_like(ret, -0.15))                   # 3 x time_penalty
    mask = torch.zeros(E, dtype=torch.uint8, device=env.device)
    mask[::2] = 1
    env.reset(mask)                                                           # plain reset: return restarts, nothing booked
    assert float(env.ep_return[::2].abs().max()) == 0 and torch.equal(env.ep_return[1::2], ret[1::2])
    assert int(env.done_count.sum()) == 0 and float(env.done_return_sum.abs().sum()) == 0
    env.step(a)
    mask.zero_()
    mask[1::2] = 1
    env.reset(mask, book_episode=True)                                        # truncation: the episode counts
    assert torch.equal(env.done_count[1::2], torch.ones(E // 2, dtype=torch.int32, device=env.device))
    assert torch.equal(env.done_steps_sum[1::2], torch.full((E // 2,), 4, dtype=torch.int32, device=env.device))
    assert torch.allclose(env.done_return_sum[1::2], torch.full((E // 2,), -0.2, device=env.device))
    assert int(env.done_count[::2].sum()) == 0 and float(env.ep_return[1::2].abs().max()) == 0
    env.reset(mask, book_episode=True)                                        # an episode with no step is not an episode
    assert int(env.done_count.sum()) == E // 2


def test_gymma_statistics_across_time_limit_truncation():
    """GymmaVecEnv truncates with gym's TimeLimit through a masked reset: every ended episode -- by the scenario
    or by the limit -- is in the statistics exactly once, and no return leaks into the next episode."""
    import torch
    from marbler_amd.gymma import GymmaVecEnv
    E, limit, T = 128, 7, 60
    env = GymmaVecEnv("robotarium_gym:PredatorCapturePrey-v0", E, time_limit=limit,
                      overrides={"predator": 3, "capture": 2, "n_agents": 5}, seed=8)
    env.reset()
    g = torch.Generator(device=env.env.device)
    g.manual_seed(1)
    ended = 0
    ret = torch.zeros(E, device=env.env.device)
    ret_sum = 0.0
    steps_sum = 0
    el = torch.zeros(E, dtype=torch.int64, device=env.env.device)
    for _ in range(T):
        r, term, _ = env.step(torch.randint(0, 5, (E, 5), generator=g, device=env.env.device, dtype=torch.int32))
        ret += r / 5                                                          # shared reward: episode return adds reward[0]
        el += 1
        ended += int(term.sum())
        ret_sum += float(ret[term].sum())
        steps_sum += int(el[term].sum())
        ret[term] = 0
        el[term] = 0
    st = env.get_stats()
    assert st["episodes"] == ended and ended >= E * (T // limit)
    assert st["steps"] == steps_sum
    assert abs(st["return_sum"] - ret_sum) < 1e-3 * max(1.0, abs(ret_sum))
    assert torch.allclose(env.env.ep_return, ret, atol=1e-5)
    env.close()


@pytest.mark.parametrize("name", ["pcp_n5_random", "warehouse_n8_random", "mt_n6_random", "simple_n6_random",
                                  "arctic_random", "pcp_n4_default"])
def test_seeded_wrapper_starts_every_episode_where_the_reference_does(name):
    """`Wrapper(seed=s)`: the state after each reset() equals the reference's (the pre-step state of the first
    step of every episode of the free-running golden vectors), float32-rounded."""
    import tempfile
    import yaml
    from marbler_amd import Wrapper
    g, scenario, cfg = load_golden(os.path.join(GOLDEN_DIR, name + ".npz"))
    seed = int(g["seeds"][0])
    per = int(g["steps_per_seed"])
    first = np.nonzero(g["first_after_reset"][:per])[0]
    if scenario == "ArcticTransport":
        first = first[:1]      # later goal columns come from Python's unseeded `random` in the reference
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(dict(cfg, seed=seed), f)
        path = f.name
    try:
        w = Wrapper(scenario, path)
    finally:
        os.unlink(path)
    if scenario == "ArcticTransport":
        import random
        w.env._pyrandom = random.Random(seed + 12345)      # how tests/golden/ref_harness.py seeded it
    for t in first:
        w.reset()
        v = w.env.vec
        assert np.array_equal(v.poses[0].cpu().numpy(), g["pre_poses"][t].astype(np.float32)), t
        if "pre_prey_loc" in g.files:
            assert np.array_equal(v.prey_loc[0].cpu().numpy(), g["pre_prey_loc"][t].astype(np.float32))
        if "pre_zone_load" in g.files:
            assert np.array_equal(v.zone_load[0].cpu().numpy(), g["pre_zone_load"][t])
        if "pre_grid" in g.files:
            assert np.array_equal(v.grid[0].cpu().numpy(), g["pre_grid"][t]) and int(v.goal_col[0]) == int(g["pre_goal_col"][t])
        assert int(v.episode_steps[0]) == 0 and float(v.carry_dist.abs().max()) == 0
    assert len(first) >= 1
    w.close()


def test_unseeded_wrappers_do_not_replay_each_other():
    """`seed: -1` = "do not seed" (every shipped YAML): two instances must not start from the same poses."""
    from marbler_amd import Wrapper
    a, b = Wrapper("Warehouse"), Wrapper("Warehouse")
    assert a.env.vec.cfg["seed"] == -1 and a.env.vec.seed != b.env.vec.seed
    a.reset()
    b.reset()
    assert not np.array_equal(a.env.vec.poses.cpu().numpy(), b.env.vec.poses.cpu().numpy())
    a.close()
    b.close()


def test_simple_files_the_violation_under_remaining():
    """simple.py:176: info['remaining'] = the violation string, and no 'message' key."""
    import tempfile
    import torch
    import yaml
    from marbler_amd import Wrapper
    from helpers import GPU_NAME, pre_state
    g, scenario, cfg = load_golden(os.path.join(GOLDEN_DIR, "viol_Simple_collision.npz"))
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(cfg, f)
        path = f.name
    try:
        w = Wrapper(scenario, path)
    finally:
        os.unlink(path)
    w.reset()
    st = pre_state(g)
    t = int(np.nonzero(g["viol"])[0][0])
    w.env.vec.load_state_dict({GPU_NAME.get(k, k): torch.as_tensor(np.asarray(v[t:t + 1])) for k, v in st.items()})
    _, rew, done, info = w.step([int(x) for x in g["actions"][t]])
    assert info["remaining"] == "collision" and "message" not in info and all(done) and rew == [-5.0] * w.n_agents
    w.close()


def test_launches_follow_torchs_current_stream_and_leave_the_current_device_alone():
    import torch
    from marbler_amd import VecRobotariumEnv
    ov = {"predator": 3, "capture": 2, "n_agents": 5}
    ref = VecRobotariumEnv("PredatorCapturePrey", 2048, overrides=ov, seed=2)
    env = VecRobotariumEnv("PredatorCapturePrey", 2048, overrides=ov, seed=2, device="cuda")   # index-less device
    assert env.device.index == torch.cuda.current_device()
    side = torch.cuda.Stream()
    g = torch.Generator(device=ref.device)
    g.manual_seed(6)
    ref.reset()
    with torch.cuda.stream(side):
        env.reset()
    for t in range(30):
        a = torch.randint(0, 5, (2048, 5), generator=g, device=ref.device, dtype=torch.int32)
        o1, r1, d1, _ = ref.step(a)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # producer and consumer ops on the side stream
            a2 = a.clone()
            o2, r2, d2, _ = env.step(a2)
            tot = o2.sum()
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and bool(torch.isfinite(tot))
    assert torch.cuda.current_device() == ref.device.index


@pytest.mark.skipif("__import__('torch').cuda.device_count() < 2")
def test_env_on_a_non_current_device():
    """A handle created for cuda:1 launches on cuda:1 while the thread's current device stays cuda:0
    (rg_* calls select the handle's device themselves)."""
    import torch
    from marbler_amd import VecRobotariumEnv
    ov = {"predator": 3, "capture": 2, "n_agents": 5}
    torch.cuda.set_device(0)
    e0 = VecRobotariumEnv("PredatorCapturePrey", 512, overrides=ov, seed=2, device="cuda:0")
    e1 = VecRobotariumEnv("PredatorCapturePrey", 512, overrides=ov, seed=2, device="cuda:1")
    e0.reset()
    e1.reset()
    for t in range(20):
        a = torch.randint(0, 5, (512, 5), dtype=torch.int32)
        o0, _, d0, _ = e0.step(a.to("cuda:0"))
        o1, _, d1, _ = e1.step(a.to("cuda:1"))
        assert torch.cuda.current_device() == 0
        assert torch.equal(o0.cpu(), o1.cpu()) and torch.equal(d0.cpu(), d1.cpu())


def test_drawn_ahead_resets_follow_a_seed_change(step_kernel):
    """The lane-group kernel draws an env's next initial state ahead of time (rg_state.next_init).  Those blocks are
    functions of (seed, env, episode): after `env.seed` changes they must not be used.  A: runs under seed 1, then
    switches to seed 2; B: a fresh env with seed 2 loaded with A's state at the switch.  From there on A == B, bit for
    bit, through many auto-resets."""
    import torch
    from marbler_amd import VecRobotariumEnv
    ov = {"predator": 3, "capture": 2, "n_agents": 5}
    E = 512
    a = VecRobotariumEnv("PredatorCapturePrey", E, overrides=ov, seed=1)
    g = torch.Generator(device=a.device)
    g.manual_seed(9)
    acts = torch.randint(0, 5, (160, E, 5), generator=g, device=a.device, dtype=torch.int32)
    a.reset()
    for t in range(40):
        a.step(acts[t])
    if step_kernel == "group":
        assert int((a.next_episode >= 0).sum()) > E // 2          # blocks are drawn ahead under seed 1
    snap = a.state_dict()
    a.seed = 2
    snap["seed"] = torch.tensor([2, 0], dtype=torch.int64)
    b = VecRobotariumEnv("PredatorCapturePrey", E, overrides=ov, seed=2)
    b.load_state_dict(snap)
    for t in range(40, 160):
        oa, ra, da, _ = a.step(acts[t])
        ob, rb, db, _ = b.step(acts[t])
        assert torch.equal(oa.view(torch.int32), ob.view(torch.int32)) and torch.equal(da, db), t
        assert torch.equal(a.poses.view(torch.int32), b.poses.view(torch.int32)), t
    assert int(a.done_count.sum()) > E and torch.equal(a.reset_count, b.reset_count)
