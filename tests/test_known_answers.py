"""Known-answer tests the reference does not have: cases whose result follows from the
reference's rules by hand, checked on the oracle (float64 and float32 tiers, CPU) and on the HIP kernels (both step
kernels, -m gpu).  Each case states the rule it pins (reference file:line)."""
import numpy as np
import pytest

from helpers import gpu_from_state

PCP2 = {"predator": 1, "capture": 1, "n_agents": 2, "num_neighbors": 1, "num_prey": 2}
BACKENDS = ["oracle_f64", "oracle_f32", pytest.param("gpu_group", marks=pytest.mark.gpu),
            pytest.param("gpu_tpe", marks=pytest.mark.gpu)]


def _cfg(scenario, ov):
    from marbler_amd.params import load_config
    return load_config(scenario, overrides=dict(ov, seed=-1))


def _step(backend, scenario, cfg, state, actions, oracle_lib, monkeypatch, steps=1):
    """state: dict of [1, ...] arrays in the oracle's naming; returns the outputs and the post-step state of env 0."""
    actions = np.asarray(actions, np.int32).reshape(1, -1)
    if backend.startswith("oracle"):
        env = oracle_lib.OracleVecEnv(scenario, cfg, 1, dtype=np.float64 if backend.endswith("64") else np.float32)
        for k, v in state.items():
            getattr(env, k)[...] = np.asarray(v).reshape(getattr(env, k).shape)
        outs = []
        for _ in range(steps):
            env.step(actions)
            outs.append({"obs": env.obs[0].astype(np.float64), "reward": env.reward[0].astype(np.float64), "done": int(env.done[0]),
                         "viol": int(env.viol[0]), "dist": env.dist[0].astype(np.float64), "remaining": int(env.remaining[0])})
        post = {k: getattr(env, k)[0].copy() for k in env.STATE_KEYS}
        return outs, post
    import torch
    from helpers import GPU_NAME
    monkeypatch.setenv("RG_STEP_KERNEL", backend.split("_")[1])
    env = gpu_from_state(scenario, cfg, state)
    outs = []
    for _ in range(steps):
        obs, rew, done, info = env.step(torch.as_tensor(actions, device=env.device))
        outs.append({"obs": obs[0].double().cpu().numpy(), "reward": rew[0].double().cpu().numpy(), "done": int(done[0]),
                     "viol": int(info["violation"][0]), "dist": info["dist_travelled"][0].double().cpu().numpy(),
                     "remaining": int(info["remaining"][0])})
    sd = env.state_dict()
    inv = {v: k for k, v in GPU_NAME.items()}
    post = {inv.get(k, k): v[0].cpu().numpy() for k, v in sd.items() if k != "seed"}
    env.close()
    return outs, post


def _pcp_state(poses, prey, sensed=None, captured=None):
    P = len(prey)
    return {"poses": np.asarray(poses, np.float64)[None], "prey_loc": np.asarray(prey, np.float64)[None],
            "prey_sensed": np.asarray(sensed if sensed is not None else [0] * P, np.uint8)[None],
            "prey_captured": np.asarray(captured if captured is not None else [0] * P, np.uint8)[None]}


@pytest.mark.parametrize("backend", BACKENDS)
def test_straight_line_drive_closed_form(backend, oracle_lib, monkeypatch):
    """One robot drives right for one env step with nothing near it (roboEnv.py:52-78, controller.py:20-24): goal
    0.2 m ahead, the projected point starts 0.05 m ahead, so |dxi| = 0.15 = the position controller's limit: v = 0.15
    for the 15 sub-steps of the first period (x += 15 * 0.033 * 0.15), then dxi = 0.15 - 0.07425 for the remaining 14.
    dist_travelled lags one sub-step (roboEnv.py:55-59): it misses the last one."""
    cfg = _cfg("PredatorCapturePrey", PCP2)
    st = _pcp_state([[0.0, -1.0], [0.0, 0.5], [0.0, 0.0]], [[1.2, -0.8], [1.2, 0.8]])
    outs, post = _step(backend, "PredatorCapturePrey", cfg, st, [1, 4], oracle_lib, monkeypatch)
    x1 = 15 * 0.033 * 0.15
    v2 = 0.15 - x1
    x2 = x1 + 14 * 0.033 * v2
    tol = 1e-12 if backend == "oracle_f64" else 2e-7
    assert abs(post["poses"][0, 0] - x2) < tol and abs(post["poses"][1, 0]) < tol and abs(post["poses"][2, 0]) < tol
    assert abs(outs[0]["dist"][0] - (x1 + 13 * 0.033 * v2)) < tol
    assert abs(post["carry"][0] - 0.033 * v2) < tol                    # the sub-step not yet counted
    # action 4: the goal is the robot's own position, but the controller steers the point 0.05 m AHEAD of the robot
    # to it, so the robot backs up: v = -0.05 for 15 sub-steps, then -(0.05 - 0.02475) for 14
    b1 = 15 * 0.033 * 0.05
    b2 = 14 * 0.033 * (0.05 - b1)
    assert abs(post["poses"][0, 1] - (-1.0 - b1 - b2)) < tol
    assert abs(outs[0]["dist"][1] - (b1 + 13 * 0.033 * (0.05 - b1))) < tol
    assert outs[0]["viol"] == 0 and outs[0]["done"] == 0
    assert abs(outs[0]["reward"][0] - cfg["time_penalty"]) < 1e-7      # nothing sensed or captured: time penalty only


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cert,radius", [("safe", 0.2), ("default", 0.17)])
def test_head_on_robots_keep_the_safety_radius(backend, cert, radius, oracle_lib, monkeypatch):
    """Two robots commanded at each other (controller.py:13-16): the barrier certificate keeps the projected points
    (0.05 m ahead of each centre) at least the safety radius apart, step after step, and nothing is flagged."""
    cfg = _cfg("PredatorCapturePrey", dict(PCP2, barrier_certificate=cert, LEFT=-1.4, RIGHT=1.4))
    st = _pcp_state([[-0.3, 0.3], [0.0, 0.0], [0.0, np.pi]], [[1.2, -0.8], [1.2, 0.8]])
    closest = 9.0
    for t in range(10):
        outs, post = _step(backend, "PredatorCapturePrey", cfg, st, [1, 0], oracle_lib, monkeypatch)
        P = post["poses"].astype(np.float64)
        xi = P[:2] + 0.05 * np.array([np.cos(P[2]), np.sin(P[2])])
        closest = min(closest, float(np.hypot(*(xi[:, 0] - xi[:, 1]))))
        assert outs[0]["viol"] == 0
        st = {"poses": P[None], "prey_loc": st["prey_loc"], "prey_sensed": post["prey_sensed"][None],
              "prey_captured": post["prey_captured"][None], "carry": post["carry"][None], "steps": post["steps"][None]}
    assert closest >= radius - 2e-3, closest          # the discrete-time barrier lets h dip by O(dt)
    assert closest < 0.3                              # ... and they did close in on each other (0.5 m apart at the start)


@pytest.mark.parametrize("backend", BACKENDS)
def test_capture_needs_sensing_and_the_no_action(backend, oracle_lib, monkeypatch):
    """PredatorCapturePrey.py:72-95: a prey is captured only once it is sensed AND a capture agent within its capture
    radius takes action 4 ('no_action'); sensing and capturing can happen in the same step; rewards 1 and 5 on top of
    the time penalty (:209-216); the episode ends when every prey is captured, info['remaining'] = 0."""
    cfg = _cfg("PredatorCapturePrey", PCP2)
    prey = [[0.3, 0.0], [1.0, 0.8]]
    poses = [[0.0, 0.35], [0.0, 0.2], [0.0, 0.0]]          # predator senses prey 0 (0.3 <= 0.45); capturer 0.206 away (<= 0.25)
    tp = cfg["time_penalty"]
    # capturer moves instead of holding: sensed, not captured
    outs, post = _step(backend, "PredatorCapturePrey", cfg, _pcp_state(poses, prey), [4, 2], oracle_lib, monkeypatch)
    assert post["prey_sensed"].tolist() == [1, 0] and post["prey_captured"].tolist() == [0, 0]
    assert abs(outs[0]["reward"][0] - (1 + tp)) < 1e-6 and outs[0]["done"] == 0
    # capturer holds (action 4): sensed and captured in one step
    outs, post = _step(backend, "PredatorCapturePrey", cfg, _pcp_state(poses, prey), [4, 4], oracle_lib, monkeypatch)
    assert post["prey_sensed"].tolist() == [1, 0] and post["prey_captured"].tolist() == [1, 0]
    assert abs(outs[0]["reward"][0] - (1 + 5 + tp)) < 1e-6 and outs[0]["done"] == 0
    # nobody senses it (the predator is far): a capturer on top of the prey captures nothing
    outs, post = _step(backend, "PredatorCapturePrey", cfg, _pcp_state([[-1.0, 0.35], [0.5, 0.2], [0.0, 0.0]], prey), [4, 4],
                       oracle_lib, monkeypatch)
    assert post["prey_sensed"].tolist() == [0, 0] and post["prey_captured"].tolist() == [0, 0]
    assert abs(outs[0]["reward"][0] - tp) < 1e-6
    # the last prey: done, remaining 0
    outs, post = _step(backend, "PredatorCapturePrey", cfg, _pcp_state(poses, prey, [0, 1], [0, 1]), [4, 4], oracle_lib, monkeypatch)
    assert post["prey_captured"].tolist() == [1, 1] and outs[0]["done"] == 1 and outs[0]["remaining"] == 0
    # observation (agent.py:19-46): the predator reports the prey it senses, the capturer (sensing radius 0) reports -5, -5
    assert np.allclose(outs[0]["obs"][0, 2:4], [-5, -5]) and np.allclose(outs[0]["obs"][1, 2:4], [-5, -5])   # captured prey are not reported


@pytest.mark.parametrize("backend", BACKENDS)
def test_nearest_neighbour_ties_go_to_the_lower_index(backend, oracle_lib, monkeypatch):
    """misc.py:20-25 leaves the order of equidistant neighbours to np.argpartition; the canonical order of this repo
    (and of the reference on this NumPy) is ascending distance, ties -> lower index."""
    cfg = _cfg("Warehouse", {"n_agents": 4, "num_neighbors": 2})
    # agents 1 and 2 sit 0.5 m above and below agent 0, agent 3 1 m to its right.  With 'no_action' every robot backs up
    # along its heading by the same arithmetic (same x, same heading), so after the step the x offsets are still
    # EXACTLY equal and |p1 - p0| == |p2 - p0| == 0.5 exactly: a true tie, in float64 and in float32
    poses = [[0.0, 0.0, 0.0, 1.0], [0.0, 0.5, -0.5, 0.0], [0.0, 0.0, 0.0, 0.0]]
    st = {"poses": np.asarray(poses, np.float64)[None], "loaded": np.zeros((1, 4), np.uint8)}
    outs, post = _step(backend, "Warehouse", cfg, st, [4, 4, 4, 4], oracle_lib, monkeypatch)
    back = 15 * 0.033 * 0.05 + 14 * 0.033 * (0.05 - 15 * 0.033 * 0.05)
    o = outs[0]["obs"][0]
    assert np.allclose(o[0:3], [-back, 0.0, 0.0], atol=1e-6)
    assert np.allclose(o[3:5], [-back, 0.5], atol=1e-6) and np.allclose(o[6:8], [-back, -0.5], atol=1e-6)   # 1 before 2
    assert post["poses"][0, 1] == post["poses"][0, 0] == post["poses"][0, 2]                                 # the tie is exact
    # agent 3 sees agent 0 first (1 m), then the tie between 1 and 2 (1.118 m) goes to agent 1
    o3 = outs[0]["obs"][3]
    assert np.allclose(o3[3:5], [-back, 0.0], atol=1e-6) and np.allclose(o3[6:8], [-back, 0.5], atol=1e-6)


@pytest.mark.parametrize("backend", BACKENDS)
def test_material_transport_zone_depletion_follows_agent_order(backend, oracle_lib, monkeypatch):
    """MaterialTransport.py:161-189: agents load in index order from a depleting zone; with 7 units left in zone 2 and
    two agents of torque 5 in it, agent 0 takes 5 and agent 1 the remaining 2; the reward adds load * load_multiplier
    per loading agent to the time penalty; an agent inside the unloading area drops its load for load * unload_multiplier."""
    cfg = _cfg("MaterialTransport", {})                   # 2 fast (torque 5) + 2 slow (torque 15)
    poses = [[1.3, 1.3, -1.3, 0.0], [-0.5, 0.5, 0.0, 0.8], [0.0, 0.0, 0.0, 0.0]]
    st = {"poses": np.asarray(poses, np.float64)[None], "load": np.array([[0, 0, 9, 0]], np.int32),
          "zone_load": np.array([[50, 7]], np.int32), "messages": np.zeros((1, 4), np.int32)}
    outs, post = _step(backend, "MaterialTransport", cfg, st, [16 + 1, 16 + 2, 16 + 3, 16], oracle_lib, monkeypatch)   # move 4 (stay), messages 1, 2, 3, 0
    assert post["load"].tolist() == [5, 2, 0, 0] and post["zone_load"].tolist() == [50, 0]
    want = cfg["time_penalty"] + 9 * cfg["unload_multiplier"] + (5 + 2) * cfg["load_multiplier"]
    assert abs(outs[0]["reward"][0] - want) < 1e-6 and np.allclose(outs[0]["reward"], outs[0]["reward"][0])
    assert post["messages"].tolist() == [1, 2, 3, 0]                                      # action % 4 (:119-120)
    # the observation is built BEFORE the loads change (:113-120 order): old loads, old zone loads, new messages
    assert outs[0]["obs"][2, 2] == 9 and outs[0]["obs"][0, 3] == 50 and outs[0]["obs"][0, 4] == 7
    assert outs[0]["obs"][0, 5:9].tolist() == [1, 2, 3, 0]


@pytest.mark.parametrize("backend", BACKENDS)
def test_driving_out_of_the_arena_is_a_boundary_violation_mid_step(backend, oracle_lib, monkeypatch):
    """rps _validate (Appendix A.4) flags x > 1.6 on the PRE-update pose of a sub-iteration; roboEnv.py:82-94 ends the step
    there: message 'boundary', reward -5, done, the violating sub-step still integrated and counted in dist_travelled."""
    cfg = _cfg("PredatorCapturePrey", dict(PCP2, RIGHT=1.75))
    st = _pcp_state([[1.56, -1.0], [0.3, -0.5], [0.0, 0.0]], [[1.2, -0.8], [1.2, 0.8]])
    outs, post = _step(backend, "PredatorCapturePrey", cfg, st, [1, 4], oracle_lib, monkeypatch)
    assert outs[0]["viol"] == 2 and outs[0]["done"] == 1 and abs(outs[0]["reward"][0] + 5) < 1e-7
    # goal = min(1.56 + 0.2, RIGHT) = 1.75, projected point at 1.61: v = 0.14.  x exceeds 1.6 after 9 sub-steps of
    # 0.033 * 0.14; the 10th validate sees it, and that sub-iteration's update still happens: 10 updates in all
    x_end = 1.56 + 10 * 0.033 * 0.14
    tol = 1e-12 if backend == "oracle_f64" else 3e-7
    assert abs(post["poses"][0, 0] - x_end) < tol
    assert abs(outs[0]["dist"][0] - 10 * 0.033 * 0.14) < tol          # on a violation the last sub-step IS counted (roboEnv.py:93)
