"""Three rows of tests/golden/ZOO_EVAL.json on the GPU, bit for bit, in both barrier-QP modes.

ZOO_EVAL.json records what the reference's own trained policies (scenarios/*/models/*.th, "the final models we evaluated
in the paper", robotarium_gym/README.md:63) score on this engine, per tier (tests/golden/make_zoo_eval.py).  The checkpoints
do not travel to the GPU box, so for three rows the float32 tier's ACTIONS of the first 64 episodes were recorded beside
their per-episode statistics (zoo_eval_replay.npz): stepping the HIP kernels from the same initial states (device sampler,
seed 2024, episode 0) with those actions must give the same episode returns, lengths, distances, violation codes and
`remaining` -- which ties the float32 rows of the record to the kernels that ship.  Both step kernels."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("step_kernel")]
REPLAY = os.path.join(GOLDEN_DIR, "zoo_eval_replay.npz")


def _rows():
    if not os.path.exists(REPLAY):
        return []
    return [str(r) for r in np.load(REPLAY)["rows"]]


@pytest.mark.parametrize("solver", ["exact", "cvxopt"])
@pytest.mark.parametrize("row", _rows())
def test_recorded_policy_rollout_reproduces_the_committed_episode_statistics(row, solver):
    """solver = cvxopt: the `float32_cvxopt_restated` rows of the record (round 5) -- the same policies on the interior-point mode,
    their own recorded actions (keys `ipm__...`)."""
    import torch
    from marbler_amd import VecRobotariumEnv
    z = np.load(REPLAY)
    key = ("ipm__" if solver == "cvxopt" else "") + row.replace("/", "__")
    acts = z[f"{key}__actions"]                       # [T, E, N] int8
    T, E, N = acts.shape
    scenario = row.split("/")[0]
    env = VecRobotariumEnv(scenario, E, seed=int(z["seed"]), auto_reset=False,
                           overrides={"barrier_solver": "cvxopt"} if solver == "cvxopt" else None)
    env.reset()
    assert env.N == N
    ret, dist = np.zeros(E, np.float64), np.zeros((E, N), np.float64)
    steps, viol, remaining = np.zeros(E, np.int64), np.zeros(E, np.int64), np.full(E, -1, np.int64)
    active = np.ones(E, bool)
    shared = bool(env.params.shared_reward)
    for j in range(T):
        _, rew, done, info = env.step(torch.as_tensor(acts[j].astype(np.int32), device=env.device))
        r = rew.cpu().numpy().astype(np.float64)
        ret[active] += (r[:, 0] if shared else r.sum(axis=1))[active]
        dist[active] += info["dist_travelled"].cpu().numpy().astype(np.float64)[active]
        ended = active & done.cpu().numpy()
        steps[ended] = j + 1
        viol[ended] = info["violation"].cpu().numpy()[ended]
        remaining[ended] = info["remaining"].cpu().numpy()[ended]
        active &= ~ended
    assert not active.any()
    assert np.array_equal(steps, z[f"{key}__steps"])
    assert np.array_equal(viol, z[f"{key}__violation"]) and np.array_equal(remaining, z[f"{key}__remaining"])
    assert np.array_equal(ret, z[f"{key}__return"]) and np.array_equal(dist, z[f"{key}__dist"])   # float64 sums of the same float32 terms
    env.close()


def test_replay_rows_are_rows_of_the_record():
    """(CPU-checkable part, but kept with the replay: the 64 recorded episodes are the first 64 of the row's float32 entry, so
    their statistics must be consistent with it -- same generator, same seed.)"""
    rows = _rows()
    assert len(rows) == 3
    rec = json.load(open(os.path.join(GOLDEN_DIR, "ZOO_EVAL.json")))
    z = np.load(REPLAY)
    for row in rows:
        for variant, prefix in (("float32", ""), ("float32_cvxopt_restated", "ipm__")):
            f32 = rec["models"][row]["variants"][variant]
            r = z[f"{prefix}{row.replace('/', '__')}__return"]
            assert abs(r.mean() - f32["return_mean"]) <= 4 * f32["return_std"] / np.sqrt(len(r)) + 1e-9, (row, variant)
