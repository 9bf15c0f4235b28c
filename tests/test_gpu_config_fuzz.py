"""Randomised configurations: scenario x agent count x neighbour slots x prey count x certificate x collision test x
penalties x controller cadence x batch size x step kernel, drawn from a seeded generator inside the ranges the parameter
block admits, each stepped free-running with auto-reset against the float32 oracle, bit for bit (the machinery of
tests/test_gpu_rollout.py).  The hand-picked cases cover the reference's configurations; this covers the combinations nobody
picked -- round 3 found by code review that rows wider than the agents fill overran the thread-per-env kernel's staging
block, a shape no listed case had."""
import numpy as np
import pytest

from test_gpu_rollout import _rollout_bit_exact, _rollout_equals_steps

pytestmark = pytest.mark.gpu


def draw_config(rng):
    scenario = str(rng.choice(["PredatorCapturePrey", "Warehouse", "MaterialTransport", "Simple", "ArcticTransport"],
                              p=[0.35, 0.2, 0.2, 0.15, 0.1]))
    ov, n_act = {}, 5
    common = {"penalize_violations": bool(rng.rand() < 0.8), "barrier_certificate": str(rng.choice(["safe", "default"], p=[0.7, 0.3])),
              "collision_variant": str(rng.choice(["offset", "center"], p=[0.7, 0.3])), "robotarium": bool(rng.rand() < 0.15)}
    if scenario == "PredatorCapturePrey":
        n = int(rng.randint(2, 17))
        npred = int(rng.randint(1, n))
        ov = {"predator": npred, "capture": n - npred, "n_agents": n, "num_prey": int(rng.choice([1, 2, 6, 8, 9, 12, 20, 33, 35])),
              "num_neighbors": int(rng.randint(0, n + 3)), "capability_aware": bool(rng.rand() < 0.4),
              "start_dist": 0.3 if n <= 16 else 0.25, "predator_radius": float(rng.choice([0.45, 0.3, 0.7])),
              "capture_radius": float(rng.choice([0.25, 0.15, 0.4]))}
        if ov["num_prey"] > 30:
            ov["step_dist"] = 0.16                      # prey grid 5 x 11 = 55 cells
    elif scenario == "Warehouse":
        n = int(rng.randint(2, 17))
        ov = {"n_agents": n, "num_neighbors": int(rng.randint(0, n + 3)), "start_dist": 0.6 if n < 12 else 0.4,
              "goal_width": float(rng.choice([0.5, 0.3, 0.9]))}
    elif scenario == "MaterialTransport":
        n = int(rng.randint(4, 17))
        nf = int(rng.randint(0, n + 1))
        ov = {"n_agents": n, "n_fast_agents": nf, "n_slow_agents": n - nf, "start_dist": 0.3 if n <= 5 else 0.25 if n <= 13 else 0.2,
              "capability_aware": bool(rng.rand() < 0.4)}
        n_act = 20
    elif scenario == "Simple":
        ov = {"n_agents": int(rng.randint(2, 17))}
    if scenario != "ArcticTransport":
        ov.update(common)
    else:
        ov.update({k: common[k] for k in ("penalize_violations", "collision_variant")})
    if ov.get("robotarium"):
        ov["update_frequency"] = int(rng.choice([10, 17]))          # a controller every sub-step: keep the oracle affordable
    E = int(rng.choice([1, 5, 63, 64, 65, 129, 300]))
    kernel = str(rng.choice(["group", "tpe"]))
    return scenario, ov, n_act, E, kernel


def draw_config_extended(rng):
    """draw_config plus the axes the first 320 draws hold fixed: sub-steps per env step (incl. a remainder chunk of every
    length and steps shorter than a controller period), episode length (resets every few steps), step length, safety
    radius, QP sweep cap and larger ragged batches.  Used by tests/fuzz_soak.py (a one-off campaign, not part of the tier)."""
    scenario, ov, n_act, E, kernel = draw_config(rng)
    if not ov.get("robotarium"):
        ov["update_frequency"] = int(rng.choice([1, 4, 5, 11, 14, 15, 16, 29, 30, 33, 44, 61, 74]))
    ov["max_episode_steps"] = int(rng.choice([1, 2, 3, 7, 20, 80]))
    if rng.rand() < 0.3:
        ov["qp_max_sweeps"] = int(rng.choice([1, 2, 3, 5, 9]))
    if scenario in ("PredatorCapturePrey", "Warehouse", "Simple") and rng.rand() < 0.5:
        ov["step_dist"] = float(rng.choice([0.16, 0.25, 0.3])) if ov.get("num_prey", 0) <= 30 else 0.16
    if rng.rand() < 0.25:
        E = int(rng.choice([257, 1000, 2049]))
    return scenario, ov, n_act, E, kernel


CASES = [draw_config(np.random.RandomState(1000 + i)) for i in range(320)]


@pytest.mark.parametrize("i", range(len(CASES)))
def test_random_configuration_is_bit_exact(i, oracle_lib, monkeypatch):
    scenario, ov, n_act, E, kernel = CASES[i]
    monkeypatch.setenv("RG_STEP_KERNEL", kernel)       # (configurations the thread-per-env kernel does not cover run the lane-group one)
    steps = 40 if scenario != "MaterialTransport" else 25
    try:
        _rollout_bit_exact(scenario, ov, n_act, steps, oracle_lib, E, require_done=False)
    except AssertionError as exc:
        raise AssertionError(f"case {i}: {scenario} {ov} E={E} kernel={kernel}: {exc}") from exc


@pytest.mark.parametrize("i", range(0, len(CASES), 4))
def test_random_configuration_multi_step_launch_equals_single_steps(i, monkeypatch):
    """The same draws through rg_rollout (K steps per launch) against K rg_step launches (GPU against GPU, every output and
    the final state)."""
    scenario, ov, n_act, E, kernel = CASES[i]     # (round 4: shapes whose E*N*D is not a multiple of 4 run too; rg_rollout takes them)
    monkeypatch.setenv("RG_STEP_KERNEL", kernel)
    _rollout_equals_steps(scenario, ov, n_act, E, K=12, reps=3, require_done=False)


def draw_barrier_family(rng):
    """draw_config plus the arguments of rps' certificate factories (config keys safety_radius, barrier_gain,
    unsafe_barrier_gain, magnitude_limit: what the reference reaches through Controller('custom', create_..._certificate2(...)),
    utilities/controller.py:11-18).  Radii above the start spacing begin inside the unsafe set (the unsafe gain at work),
    magnitude limits below the position controller's 0.15 exercise the 'threshold control inputs' branch."""
    scenario, ov, n_act, E, kernel = draw_config(rng)
    ov["safety_radius"] = float(rng.choice([0.12, 0.17, 0.2, 0.25, 0.32]))
    ov["barrier_gain"] = float(rng.choice([10.0, 100.0, 1000.0]))
    ov["unsafe_barrier_gain"] = float(rng.choice([1e4, 1e6, 1e7]))
    ov["magnitude_limit"] = float(rng.choice([0.1, 0.15, 0.2, 0.3]))
    return scenario, ov, n_act, E, kernel


BARRIER_CASES = [draw_barrier_family(np.random.RandomState(5000 + i)) for i in range(48)]


@pytest.mark.parametrize("i", range(len(BARRIER_CASES)))
def test_random_barrier_family_is_bit_exact(i, oracle_lib, monkeypatch):
    scenario, ov, n_act, E, kernel = BARRIER_CASES[i]
    monkeypatch.setenv("RG_STEP_KERNEL", kernel)
    steps = 40 if scenario != "MaterialTransport" else 25
    try:
        _rollout_bit_exact(scenario, ov, n_act, steps, oracle_lib, E, require_done=False)
    except AssertionError as exc:
        raise AssertionError(f"case {i}: {scenario} {ov} E={E} kernel={kernel}: {exc}") from exc


def draw_interior_point(rng):
    """draw_config restricted to what `barrier_solver: cvxopt` admits (n_agents <= 8), with the certificate family and cvxopt's own
    options drawn too (tight tolerances: many iterations; a small iteration cap: the cap binds)."""
    while True:
        scenario, ov, n_act, E, kernel = draw_barrier_family(rng) if rng.rand() < 0.5 else draw_config(rng)
        n = int(ov.get("n_agents", 4))
        if n <= 8:
            break
    ov["barrier_solver"] = "cvxopt"
    if rng.rand() < 0.3:
        ov["cvxopt_reltol"] = float(rng.choice([1e-1, 1e-3, 1e-5]))
        ov["cvxopt_feastol"] = float(rng.choice([1e-1, 1e-2, 1e-4]))
    if rng.rand() < 0.2:
        ov["cvxopt_maxiters"] = int(rng.choice([0, 1, 4, 12]))
    return scenario, ov, n_act, min(E, 129), kernel


IPM_CASES = [draw_interior_point(np.random.RandomState(7000 + i)) for i in range(64)]


@pytest.mark.parametrize("i", range(len(IPM_CASES)))
def test_random_interior_point_configuration_is_bit_exact(i, oracle_lib, monkeypatch):
    scenario, ov, n_act, E, kernel = IPM_CASES[i]
    monkeypatch.setenv("RG_STEP_KERNEL", kernel)
    steps = 30 if scenario != "MaterialTransport" else 16
    try:
        _rollout_bit_exact(scenario, ov, n_act, steps, oracle_lib, E, require_done=False)
    except AssertionError as exc:
        raise AssertionError(f"case {i}: {scenario} {ov} E={E} kernel={kernel}: {exc}") from exc
