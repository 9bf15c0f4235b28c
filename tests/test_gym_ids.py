"""The EPyMARL entry point: `env_args.key="robotarium_gym:<Scenario>-v0"` (README.md:26 of the reference).
Importing `robotarium_gym` must register the reference's five ids with the reference's entry-point string and
kwargs (/root/reference/robotarium_gym/__init__.py:4-23); `gym.make(key)` must then build a gym.Env that
TimeLimit can wrap and that steps.  gym is not installed in this image: the child processes run with
tests/stub_gym (a recording stand-in for gym's registration / make / TimeLimit) first on sys.path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "stub_gym")
IDS = ["PredatorCapturePrey-v0", "Warehouse-v0", "Simple-v0", "ArcticTransport-v0", "MaterialTransport-v0"]


def _child(code):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STUB, ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_importing_robotarium_gym_registers_the_reference_ids():
    out = _child("""
import json, os, importlib
import gym
from gym.envs.registration import CALLS
import robotarium_gym
calls = [(i, k["entry_point"], sorted(k["kwargs"]), k["kwargs"]["env_name"], os.path.exists(k["kwargs"]["config_path"]),
          sorted(k)) for i, k in CALLS]
mod, attr = CALLS[0][1]["entry_point"].split(":")
cls = getattr(importlib.import_module(mod), attr)
print(json.dumps({"calls": calls, "is_env": issubclass(cls, gym.Env), "cls": cls.__module__ + "." + cls.__name__,
                  "registered": robotarium_gym.REGISTERED}))
""")
    assert [c[0] for c in out["calls"]] == IDS                      # the reference's ids, in its order
    for gym_id, entry, kw, env_name, cfg_exists, reg_keys in out["calls"]:
        assert entry == "robotarium_gym.wrapper:Wrapper"
        assert kw == ["config_path", "env_name"] and reg_keys == ["entry_point", "kwargs"]
        assert env_name + "-v0" == gym_id and cfg_exists
    assert out["is_env"], "Wrapper must derive from gym.Env when gym is importable (reference wrapper.py:19)"
    assert out["registered"] == IDS


def test_without_gym_the_package_still_imports():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.environ.get("PYTHONPATH", "")]))
    code = "import robotarium_gym, marbler_amd.wrapper as w; print(robotarium_gym.REGISTERED, w.Wrapper.__mro__[1].__name__)"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    try:
        import gym  # noqa: F401
    except ImportError:
        try:
            import gymnasium  # noqa: F401
        except ImportError:
            assert r.stdout.split()[0] == "[]" and r.stdout.split()[-1] == "object"


@pytest.mark.gpu
@pytest.mark.parametrize("gym_id", IDS)
def test_gym_make_builds_and_steps_the_registered_entry_point(gym_id):
    out = _child(f"""
import json
import numpy as np
import gym
from gym.wrappers import TimeLimit
env = TimeLimit(gym.make("robotarium_gym:{gym_id}"), max_episode_steps=4)   # what EPyMARL's gymma wrapper does
assert isinstance(env.unwrapped, gym.Env) and env.unwrapped.spec.id == "{gym_id}"
obs = env.reset()
n = env.unwrapped.n_agents
assert len(obs) == n and len(env.action_space) == n and len(env.observation_space) == n
moved = 0.0
for t in range(4):
    obs, rew, done, info = env.step([a.sample() for a in env.action_space])
    assert isinstance(obs, tuple) and len(obs) == n and len(rew) == n and len(done) == n
    moved += float(np.sum(info["dist_travelled"]))
print(json.dumps({{"n": n, "done": bool(all(done)), "truncated": bool(info.get("TimeLimit.truncated", False)),
                  "obs_dim": int(len(obs[0])), "space_dim": int(env.observation_space[0].shape[0]), "moved": moved}}))
env.close()
""")
    assert out["done"] and out["obs_dim"] == out["space_dim"] and out["moved"] > 0.0
