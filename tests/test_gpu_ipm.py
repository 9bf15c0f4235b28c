"""`barrier_solver: cvxopt` (rg_scenario_params.qp_mode = RG_QP_CVXOPT): the barrier certificate's QP as the reference's stack
evaluates it -- rps hands it to cvxopt's interior-point `qp` at reltol = feastol = 1e-2, maxiters 50 (utilities/controller.py:13-16,23)
and gets an approximate, strictly interior iterate, not the projection.  The HIP kernels run the restated iteration in binary64
(csrc/ipm_qp.h); here they are held, bit for bit, against the float tier of the CPU oracle (oracle_core.h barrier_qp_ipm_spec) in
free-running rollouts with auto-reset, for every agent count the mode admits (1 .. 8), all five scenarios, ragged batches, the
gymma block and the multi-step launch.  (What the float tier itself is worth against the float64 restatement of cvxopt and the
reference's own Python: tests/test_ipm_spec.py, tests/test_oracle_golden.py -- CPU tier.)"""
import numpy as np
import pytest

from test_gpu_rollout import _rollout_bit_exact, _rollout_equals_steps

pytestmark = pytest.mark.gpu
IPM = {"barrier_solver": "cvxopt"}

CASES = [("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 170, 192),        # BASELINE configs[1]'s shape
         ("PredatorCapturePrey", {}, 5, 120, 65),                                                   # the reference's default: 4 agents
         ("Warehouse", {"n_agents": 8}, 5, 120, 100),                                               # configs[2]: 16 unknowns, 28 rows
         ("Warehouse", {}, 5, 110, 64),                                                             # 6 agents
         ("Warehouse", {"n_agents": 7, "barrier_certificate": "default"}, 5, 110, 33),              # certificate (no unsafe gain), radius 0.17
         ("MaterialTransport", {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}, 20, 80, 64),   # configs[4]
         ("MaterialTransport", {}, 20, 80, 64),
         ("Simple", {}, 5, 100, 64),
         ("Simple", {"n_agents": 2}, 5, 130, 7),
         ("ArcticTransport", {}, 5, 120, 65),
         ("PredatorCapturePrey", {"predator": 1, "capture": 1, "n_agents": 2, "num_neighbors": 0}, 5, 150, 1),
         ("PredatorCapturePrey", {"predator": 2, "capture": 1, "n_agents": 3, "num_prey": 1, "num_neighbors": 1}, 5, 120, 33),
         ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "robotarium": True, "update_frequency": 10}, 5, 60, 40),  # a QP every sub-step
         ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "safety_radius": 0.3, "barrier_gain": 1000.0,
                                  "magnitude_limit": 0.1}, 5, 100, 64),                             # stiff: starts inside the unsafe set
         ("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5, "cvxopt_maxiters": 3}, 5, 80, 64)]   # the iteration cap binds


@pytest.mark.parametrize("kernel", ["group", "tpe"])
@pytest.mark.parametrize("scenario,ov,n_act,steps,E", CASES)
def test_interior_point_mode_rollout_is_bit_exact(scenario, ov, n_act, steps, E, kernel, oracle_lib, monkeypatch):
    """Both step kernels: a lane group per env (rows in LDS, row phases spread over the group's lanes; every agent count up to 8)
    and one lane per env (everything in the lane's registers; up to 5 agents -- the library falls back to the lane-group kernel
    above that, so those cases run it twice)."""
    import torch
    from marbler_amd import VecRobotariumEnv
    monkeypatch.setenv("RG_STEP_KERNEL", kernel)
    probe = VecRobotariumEnv(scenario, 4, overrides=dict(ov, **IPM))
    n, which = probe.N, probe.step_kernel
    probe.close()
    assert which == (kernel if (kernel == "group" or 2 <= n <= 5) else "group"), (n, which)
    _rollout_bit_exact(scenario, dict(ov, **IPM), n_act, steps, oracle_lib, E, require_done=False)


@pytest.mark.parametrize("scenario,ov,n_act,E", [("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5, 130),
                                                  ("Warehouse", {"n_agents": 8}, 5, 65),
                                                  ("MaterialTransport", {}, 20, 64), ("ArcticTransport", {}, 5, 33)])
@pytest.mark.parametrize("kernel", ["group", "tpe"])
def test_interior_point_mode_multi_step_launch_equals_single_steps(scenario, ov, n_act, E, kernel, monkeypatch):
    monkeypatch.setenv("RG_STEP_KERNEL", kernel)
    _rollout_equals_steps(scenario, dict(ov, **IPM), n_act, E, K=10, reps=3, require_done=False)


def test_interior_point_mode_picks_the_kernel_by_batch_size():
    """The library's own choice (no RG_STEP_KERNEL): a lane group per env up to the cross-over, one lane per env beyond, and the
    two agree bit for bit (a short free-running comparison at a batch on either side is in the parametrised cases above)."""
    from marbler_amd import VecRobotariumEnv
    ov = dict({"predator": 3, "capture": 2, "n_agents": 5}, **IPM)
    for E, want in ((4096, "group"), (8192, "group"), (16384, "tpe"), (65536, "tpe")):
        env = VecRobotariumEnv("PredatorCapturePrey", E, overrides=ov)
        assert env.step_kernel == want, (E, env.step_kernel)
        env.close()
    env = VecRobotariumEnv("Warehouse", 65536, overrides=dict({"n_agents": 8}, **IPM))
    assert env.step_kernel == "group"          # N > 5: the lane-group kernel at every batch size
    env.close()


def test_interior_point_mode_is_refused_above_eight_agents():
    from marbler_amd import load_config, make_params
    with pytest.raises(ValueError, match="n_agents <= 8"):
        make_params("Warehouse", load_config("Warehouse", overrides=dict(IPM, n_agents=9, start_dist=0.4)))


def test_interior_point_mode_differs_from_the_projection_and_stays_inside():
    """The two solvers are different functions: from the same states the interior-point iterate leaves the robots a little further
    apart (it stops strictly inside the feasible set), so over a random-policy rollout it produces fewer collisions."""
    import torch
    from marbler_amd import VecRobotariumEnv
    E, ov = 2048, {"predator": 3, "capture": 2, "n_agents": 5}
    counts = {}
    for solver in ("exact", "cvxopt"):
        env = VecRobotariumEnv("PredatorCapturePrey", E, overrides=dict(ov, barrier_solver=solver), seed=11, auto_reset=True)
        env.reset()
        g = torch.Generator(device=env.device).manual_seed(4)
        n = 0
        for t in range(160):
            _, _, _, info = env.step(torch.randint(0, 5, (E, 5), generator=g, device=env.device, dtype=torch.int32))
            n += int(((info["violation"] & 1) != 0).sum())
        counts[solver] = n
        env.close()
    assert counts["exact"] > 0 and counts["cvxopt"] < counts["exact"], counts
