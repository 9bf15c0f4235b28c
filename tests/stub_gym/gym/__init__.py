"""TEST STAND-IN for the `gym` package (absent from this image), only ever put on sys.path by
tests/test_gym_ids.py in a child process.  It mimics what `gym.make("<module>:<id>")` does with a registered
id -- import the module named before the colon (registration is an import side effect), look the id up,
import the entry point, construct it with the registered kwargs, then `env.unwrapped.spec = spec` -- and the
attributes gym's wrappers read from an Env (metadata, reward_range, spec, unwrapped)."""
import importlib

from . import spaces  # noqa: F401
from .envs.registration import REGISTRY, EnvSpec, register  # noqa: F401


class Env(object):
    metadata = {"render.modes": []}
    reward_range = (-float("inf"), float("inf"))
    spec = None
    action_space = None
    observation_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        pass

    def seed(self, seed=None):
        return

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space
        self.reward_range = env.reward_range
        self.metadata = env.metadata

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def spec(self):
        return self.env.spec

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def close(self):
        return self.env.close()


def make(id, **kwargs):
    if ":" in id:
        mod, id = id.split(":", 1)
        importlib.import_module(mod)
    spec = REGISTRY[id]
    mod_name, attr = spec.entry_point.split(":")
    cls = getattr(importlib.import_module(mod_name), attr)
    kw = dict(spec.kwargs)
    kw.update(kwargs)
    env = cls(**kw)
    env.unwrapped.spec = spec
    return env
