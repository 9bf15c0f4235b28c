"""The batched actor (marbler_amd/evaluate.py) against the reference's own RNNAgent / RNNNSAgent
modules (golden vectors from tests/golden/make_actor_golden.py), on the CPU; and the device
evaluation loop on the GPU."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN_DIR, "actor_*.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_batched_actor_matches_reference_modules(path):
    from marbler_amd.evaluate import BatchedActor
    g = np.load(path)
    sd = {k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("sd_")}
    T, N, I = g["inputs"].shape
    actor = BatchedActor(sd, N, use_rnn=bool(g["use_rnn"]))
    E = 3                                            # the same env three times: batching must not mix rows
    h = actor.init_hidden(E)
    for t in range(T):
        x = torch.as_tensor(g["inputs"][t]).unsqueeze(0).expand(E, N, I).contiguous()
        q, h = actor.forward(x, h)
        for e in range(E):
            assert np.abs(q[e].numpy() - g["q"][t]).max() < 1e-5
            assert np.abs(h[e].numpy() - g["h"][t]).max() < 1e-5


@pytest.mark.gpu
def test_device_evaluation_loop():
    from marbler_amd import VecRobotariumEnv
    from marbler_amd.evaluate import BatchedActor, run_eval
    g = np.load(os.path.join(GOLDEN_DIR, "actor_shared_gru.npz"))
    sd = {k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("sd_")}
    env = VecRobotariumEnv("PredatorCapturePrey", 256, seed=5)        # default config: N = 4, D = 16 (+4 ids = 20)
    actor = BatchedActor(sd, env.N, device=env.device)
    out = run_eval(env, actor, steps=120)
    assert out["episodes"] >= 256 and np.isfinite(out["mean_return"]) and 0 < out["mean_steps"] <= 81
    # deterministic: the same seed gives the same statistics
    env2 = VecRobotariumEnv("PredatorCapturePrey", 256, seed=5)
    assert run_eval(env2, actor, steps=120) == out
    with pytest.raises(ValueError):
        run_eval(env2, actor, steps=1, obs_agent_id=False)
    # one hipGraph launch per step instead of a dozen kernel launches: same statistics
    env3 = VecRobotariumEnv("PredatorCapturePrey", 256, seed=5)
    assert run_eval(env3, actor, steps=120, use_graph=True) == out
    # the env is back on its own stream and still steps
    obs, _, _, _ = env3.step(torch.zeros(256, env3.N, dtype=torch.int32, device=env3.device))
    assert torch.isfinite(obs).all()


@pytest.mark.gpu
def test_evaluate_cli_on_a_model_zoo_shaped_checkpoint(tmp_path):
    """`python -m marbler_amd.evaluate` (the batched `python -m robotarium_gym.main`): a `.th` state dict and a
    sacred `.json` in the reference's formats (random weights of the zoo's shapes; the zoo itself stays in
    the reference), default PredatorCapturePrey config (4 agents, 16 + 4 inputs)."""
    import json
    from marbler_amd import evaluate
    from test_gpu_actor import _random_actor
    sd = _random_actor(1, 20, 64, 5, True, seed=11)
    torch.save(sd, tmp_path / "qmix.th")
    (tmp_path / "qmix.json").write_text(json.dumps({"use_rnn": True, "obs_agent_id": True, "hidden_dim": 64, "agent": "rnn"}))
    argv = ["--scenario", "PredatorCapturePrey", "--model-file", str(tmp_path / "qmix.th"),
            "--model-config", str(tmp_path / "qmix.json"), "--envs", "96", "--steps", "100"]
    fused = evaluate.main(argv)
    eager = evaluate.main(argv + ["--torch-actor"])
    assert fused["episodes"] >= 96 and 0 < fused["mean_steps"] <= 81 and np.isfinite(fused["mean_return"])
    # same policy, float32 both ways: the two evaluations may part ways only through 1e-6-level ties in arg-max
    assert abs(fused["mean_return"] - eager["mean_return"]) < 1.0 and abs(fused["episodes"] - eager["episodes"]) <= 10


ZOO = "/root/reference/robotarium_gym/scenarios"


@pytest.mark.skipif(not os.path.isdir(ZOO), reason="the reference tree (with its model zoo) is not on this machine")
def test_every_model_of_the_reference_zoo_loads_and_fits_its_scenario():
    """misc.py:65-91 without the env: each `.th` / `.json` pair of the reference's model zoo loads into
    BatchedActor (shared and non-shared, GRU and MLP), its input width is the scenario's observation
    width (+ agent id) under this package's copy of that scenario's config, its action count is the
    scenario's, and its shape is one the fused kernel takes.  (Read here only; nothing of the zoo is
    copied into the repo.)"""
    import json
    from marbler_amd.evaluate import load_actor
    from marbler_amd.gymma import N_ACTIONS
    from marbler_amd.params import load_config, make_params
    seen = 0
    for scenario in sorted(os.listdir(ZOO)):
        mdir = os.path.join(ZOO, scenario, "models")
        if scenario not in N_ACTIONS or not os.path.isdir(mdir):
            continue
        p = make_params(scenario, load_config(scenario))
        for js in sorted(glob.glob(os.path.join(mdir, "*.json"))):
            th = js[:-5] + ".th"
            if not os.path.exists(th):
                continue
            actor, cfg = load_actor(th, js, p.n_agents, device="cpu")
            width = p.obs_dim + (p.n_agents if cfg.get("obs_agent_id", True) else 0)
            assert actor.input_dim == width, (scenario, os.path.basename(js), actor.input_dim, width)
            assert actor.n_actions == N_ACTIONS[scenario], (scenario, os.path.basename(js))
            assert actor.hidden_dim in (64, 128) and actor.n_actions <= 32 and actor.input_dim <= 64
            assert actor.non_shared == (cfg.get("agent") == "rnn_ns")
            q, h = actor.forward(torch.zeros(2, p.n_agents, width), actor.init_hidden(2))
            assert tuple(q.shape) == (2, p.n_agents, actor.n_actions) and torch.isfinite(q).all()
            seen += 1
    assert seen >= 15
