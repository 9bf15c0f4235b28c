"""BASELINE.json's batch shapes and every dispatch width of the lane-group kernel against the float32 oracle.

The lane-group launch picks the number of env slots a wavefront uses from the batch size (`launch_step_scn`,
csrc/step_group.h): at 8 lanes per env a batch of <= 1024 envs runs 1 env per wave, 2048 -> 2, 4096 -> 4 (the
headline: 1024 workgroups through the XCD-aware chunk map), >= 8192 -> 8; at 4 lanes per env the steps are 1 / 2 / 4 /
8 / 16.  The other oracle-checked rollouts (tests/test_gpu_rollout.py) stay below 1024 envs = one env per wave, so
these cases put the lane -> env map the driver's bench line times -- and the per-GPU shapes of configs[2] .. [4] --
under the same bar: free-running rollouts with auto-reset, every output and state word of every step bit for bit
against the oracle stepping the same envs on the host with the reset twin, through rg_step and through rg_rollout.

Reference rows: PredatorCapturePrey.py:138-176, warehouse.py:102-122, MaterialTransport.py:113-148.
"""
import os

import numpy as np
import pytest

from helpers import oracle_reset, oracle_reset_params

pytestmark = pytest.mark.gpu

PCP5 = {"predator": 3, "capture": 2, "n_agents": 5}
WH8 = {"n_agents": 8}
MT6 = {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}

# (id, scenario, overrides, action count, envs, steps, kernel, env slots per wave the lane-group dispatch must pick)
CASES = [
    ("pcp-4096x5-headline", "PredatorCapturePrey", PCP5, 5, 4096, 200, "group", 4),   # configs[1]: 1024 blocks, xcd_chunk
    ("pcp-2048x5", "PredatorCapturePrey", PCP5, 5, 2048, 120, "group", 2),
    ("pcp-8192x5", "PredatorCapturePrey", PCP5, 5, 8192, 120, "group", 8),
    ("pcp-32768x5", "PredatorCapturePrey", PCP5, 5, 32768, 90, "group", 8),           # configs[3] on one GPU
    ("pcp-4095x5-ragged", "PredatorCapturePrey", PCP5, 5, 4095, 100, "group", 4),
    ("pcp-2047x5-ragged", "PredatorCapturePrey", PCP5, 5, 2047, 100, "group", 2),
    ("pcp-8191x5-ragged", "PredatorCapturePrey", PCP5, 5, 8191, 100, "group", 8),
    ("pcp-1025x5-ragged", "PredatorCapturePrey", PCP5, 5, 1025, 100, "group", 2),
    ("warehouse-4096x8", "Warehouse", WH8, 5, 4096, 120, "group", 4),                  # configs[2]
    ("warehouse-8193x8-ragged", "Warehouse", WH8, 5, 8193, 110, "group", 8),
    ("mt-2048x6", "MaterialTransport", MT6, 20, 2048, 100, "group", 2),                # configs[4], per-GPU share
    ("mt-4096x6", "MaterialTransport", MT6, 20, 4096, 80, "group", 4),
    ("mt-4096x4-default", "MaterialTransport", {}, 20, 4096, 80, "group", 4),         # 4 lanes per env: 16 slots, 4 used
    ("pcp-4096x4-default", "PredatorCapturePrey", {}, 5, 4096, 100, "group", 4),
    ("arctic-16384x4", "ArcticTransport", {}, 5, 16384, 60, "group", None),           # fixed 16 envs per wave
    ("pcp-4096x5-tpe", "PredatorCapturePrey", PCP5, 5, 4096, 100, "tpe", None),        # thread-per-env, forced
    ("pcp-65536x5-auto", "PredatorCapturePrey", PCP5, 5, 65536, 40, None, None),      # the library picks thread-per-env
    ("mt-4094x6-tpe", "MaterialTransport", MT6, 20, 4094, 80, "tpe", None),             # ragged; E*N*D % 4 == 0 for rg_rollout
]


def expected_slots(N, E):
    """launch_step_scn's rule, restated: halve the env slots per wave while the batch still fits 1024 waves."""
    gw = 4 if N <= 4 else 8 if N <= 8 else 16
    epw = 64 // gw
    while epw >= 2 and (E + epw // 2 - 1) // (epw // 2) <= 1024:
        epw //= 2
    return epw


@pytest.mark.parametrize("name,scenario,ov,n_act,E,steps,kernel,slots", CASES, ids=[c[0] for c in CASES])
def test_baseline_shape_bit_exact_vs_oracle(name, scenario, ov, n_act, E, steps, kernel, slots, oracle_lib, monkeypatch):
    if kernel is None:
        monkeypatch.delenv("RG_STEP_KERNEL", raising=False)
    else:
        monkeypatch.setenv("RG_STEP_KERNEL", kernel)
    shape_rollout_vs_oracle(name, scenario, ov, n_act, E, steps, slots, oracle_lib)


def shape_rollout_vs_oracle(name, scenario, ov, n_act, E, steps, slots, oracle_lib, check_rollout=True, seed=1234, action_seed=17):
    """(RG_STEP_KERNEL is the caller's business.)  check_rollout=False: rg_step against the oracle only, no [K, ...] arrays
    (tests/soak_shapes.py runs thousands of steps this way)."""
    import torch
    from marbler_amd import VecRobotariumEnv
    threads = max(1, min(16, (os.cpu_count() or 1)))
    env = VecRobotariumEnv(scenario, E, overrides=ov, seed=seed, auto_reset=True, collect_qp_stats=True)
    twin = VecRobotariumEnv(scenario, E, overrides=ov, seed=seed, auto_reset=True, collect_qp_stats=True) if check_rollout else None  # rg_rollout
    N = env.N
    if slots is not None:
        assert expected_slots(N, E) == slots, "the case no longer exercises the dispatch width it is named for"
    cfg = dict(env.cfg)
    orc = oracle_lib.OracleVecEnv(scenario, cfg, E, dtype=np.float32)
    rp = oracle_reset_params(oracle_lib, env.params)
    env.reset()
    if twin is not None:
        twin.reset()
    for e in range(E):
        oracle_reset(oracle_lib, orc, rp, seed, e, 0)
    assert np.array_equal(env.poses.cpu().numpy().view(np.uint32), orc.poses.view(np.uint32))
    episodes = np.zeros(E, np.int64)
    rng = np.random.RandomState(action_seed)
    acts = rng.randint(0, n_act, size=(steps, E, N)).astype(np.int32) if check_rollout else None
    acts_dev = torch.as_tensor(acts, device=env.device) if check_rollout else None
    kept = {k: [] for k in ("obs", "reward", "done", "dist_travelled", "violation", "remaining", "qp_sweeps")}
    n_done = n_viol = 0
    for t in range(steps):
        act_t = acts[t] if check_rollout else rng.randint(0, n_act, size=(E, N)).astype(np.int32)
        obs, rew, done, info = env.step(acts_dev[t] if check_rollout else torch.as_tensor(act_t, device=env.device))
        o_obs, o_rew, o_done, o_info = orc.step(act_t, threads=threads)
        got = {"obs": obs, "reward": rew, "done": env.done_u8, "dist_travelled": info["dist_travelled"],
               "violation": info["violation"], "remaining": info["remaining"], "qp_sweeps": env.qp_sweeps}
        if check_rollout:
            for k_, v in got.items():
                kept[k_].append(v.clone())
        host = {k_: v.cpu().numpy() for k_, v in got.items()}

        def same(a, b, what):
            if a.dtype == np.float32:
                a, b = a.view(np.uint32), b.view(np.uint32)
            if not np.array_equal(a, b):
                bad = np.nonzero((a != b).reshape(E, -1).any(axis=1))[0]
                raise AssertionError(f"{name}: {what} differs at step {t} in {len(bad)} envs, first {bad[:8].tolist()}")

        same(host["done"], o_done, "done")
        same(host["violation"], o_info["violation"], "violation")
        same(host["remaining"], o_info["remaining"], "remaining")
        same(host["obs"], o_obs, "obs")
        same(host["reward"], o_rew, "reward")
        same(host["dist_travelled"], o_info["dist_travelled"], "dist_travelled")
        same(host["qp_sweeps"], orc.qp_sweeps, "qp_sweeps")
        for e in np.nonzero(o_done)[0]:
            episodes[e] += 1
            oracle_reset(oracle_lib, orc, rp, seed, e, int(episodes[e]))
        n_done += int(o_done.sum())
        n_viol += int((o_info["violation"] > 0).sum())
        # the state after the (possibly reset) step
        same(env.poses.cpu().numpy(), orc.poses, "poses")
        same(env.carry_dist.cpu().numpy(), orc.carry, "carry_dist")
        same(env.episode_steps.cpu().numpy(), orc.steps, "episode_steps")
        if scenario == "PredatorCapturePrey":
            same(env.prey_loc.cpu().numpy(), orc.prey_loc, "prey_loc")
            same(env.prey_sensed.cpu().numpy(), orc.prey_sensed, "prey_sensed")
            same(env.prey_captured.cpu().numpy(), orc.prey_captured, "prey_captured")
        elif scenario == "Warehouse":
            same(env.loaded.cpu().numpy(), orc.loaded, "loaded")
        elif scenario == "MaterialTransport":
            same(env.load.cpu().numpy(), orc.load, "load")
            same(env.zone_load.cpu().numpy(), orc.zone_load, "zone_load")
            same(env.messages.cpu().numpy(), orc.messages, "messages")
        elif scenario == "ArcticTransport":
            same(env.grid.cpu().numpy(), orc.grid, "grid")
            same(env.reached_goal.cpu().numpy(), orc.reached_goal, "reached_goal")
    assert n_done > 0 and n_viol > 0, "the rollout must contain episode ends and violations"
    assert np.array_equal(env.done_count.cpu().numpy(), episodes)
    assert np.array_equal(env.reset_count.cpu().numpy(), episodes + 1)   # the explicit reset() drew episode 0
    if not check_rollout:
        env.close()
        return {"env_steps": steps * E, "episodes": int(episodes.sum()), "violations": n_viol}
    # ---- the same action sequence through rg_rollout (one launch for all steps): what the oracle just confirmed, step by step
    out = twin.rollout(acts_dev)
    for k_ in kept:
        ref = torch.stack(kept[k_])
        assert torch.equal(out[k_].view(ref.dtype) if out[k_].dtype != ref.dtype else out[k_], ref), f"{name}: rg_rollout {k_}"
    sa, sb = env.state_dict(), twin.state_dict()
    for key in sa:
        assert torch.equal(sa[key], sb[key]), f"{name}: rg_rollout state {key}"
    env.close()
    twin.close()


@pytest.mark.parametrize("scenario,ov,threshold", [
    ("PredatorCapturePrey", PCP5, 65536),
    ("PredatorCapturePrey", {"predator": 2, "capture": 1, "n_agents": 3}, 98304),
    ("PredatorCapturePrey", {}, 196608),                                                   # the reference's default: 4 agents
    ("Simple", {}, 393216),
    ("MaterialTransport", MT6, 65536),
    ("MaterialTransport", {"n_agents": 5, "n_fast_agents": 3, "n_slow_agents": 2, "start_dist": 0.25}, 49152),
    ("Warehouse", {}, 262144),                                                             # default: 6 agents
    ("PredatorCapturePrey", {"predator": 3, "capture": 3, "n_agents": 6}, 98304),
    ("Warehouse", WH8, None),                                                              # N >= 7: never
])
def test_kernel_choice_follows_the_measured_cross_overs(scenario, ov, threshold, monkeypatch):
    """`rg_create` picks the step kernel from (scenario, agent count, batch size): the table of robogym_capi.hip
    `tpe_min_envs`, measured with tools/crossover_probe.py (profiles/r3_crossover_probe.txt).  `rg_step_kernel` reports it."""
    from marbler_amd import VecRobotariumEnv
    monkeypatch.delenv("RG_STEP_KERNEL", raising=False)
    for E, want in ((4096, "group"),) + (((threshold - 1, "group"), (threshold, "tpe")) if threshold else ((524288, "group"),)):
        env = VecRobotariumEnv(scenario, E, overrides=ov, seed=0)
        assert env.step_kernel == want, (scenario, E, env.step_kernel)
        env.close()
    monkeypatch.setenv("RG_STEP_KERNEL", "tpe")
    env = VecRobotariumEnv(scenario, 64, overrides=ov, seed=0)
    assert env.step_kernel == ("tpe" if env.N <= 6 else "group")   # N <= 6 has a thread-per-env instantiation; above, the lane-group kernel
    env.close()
