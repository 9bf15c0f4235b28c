"""Observation / action space descriptions.  gym (or gymnasium) spaces when one is installed --
the reference builds `gym.spaces` objects (PredatorCapturePrey.py:47-56) -- otherwise small
stand-ins with the same attributes, so the package imports without gym."""
import numpy as np

try:  # pragma: no cover - depends on the environment
    from gym import spaces as _spaces
    HAVE_GYM = "gym"
except Exception:  # noqa: BLE001
    try:
        from gymnasium import spaces as _spaces
        HAVE_GYM = "gymnasium"
    except Exception:  # noqa: BLE001
        _spaces = None
        HAVE_GYM = None


class _Discrete(object):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def sample(self):
        return int(np.random.randint(self.n))

    def __repr__(self):
        return f"Discrete({self.n})"


class _Box(object):
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {np.dtype(self.dtype).name})"


class _Tuple(object):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def sample(self):
        return tuple(s.sample() for s in self.spaces)

    def __repr__(self):
        return f"Tuple({', '.join(map(repr, self.spaces))})"


Discrete = _spaces.Discrete if _spaces else _Discrete
Box = _spaces.Box if _spaces else _Box
Tuple = _spaces.Tuple if _spaces else _Tuple


def scenario_spaces(scenario, params):
    """(action_space, observation_space) exactly as the scenario constructors build them:
    PredatorCapturePrey.py:47-56 (Discrete(5), Box(-5, 3)), warehouse.py:68-77 (Discrete(5),
    Box(-1.5, 1.5)), MaterialTransport.py:79-87 (Discrete(20), Box(-1.5, 1.5))."""
    N, D = params.n_agents, params.obs_dim
    if scenario == "PredatorCapturePrey":
        n_act, lo, hi = 5, -5, 3
    elif scenario == "Warehouse":
        n_act, lo, hi = 5, -1.5, 1.5
    elif scenario in ("Simple", "ArcticTransport"):     # simple.py:96-99, ArcticTransport.py:41-47
        n_act, lo, hi = 5, -1.5, 3
    else:
        n_act, lo, hi = 20, -1.5, 1.5
    actions = Tuple(tuple(Discrete(n_act) for _ in range(N)))
    observations = Tuple(tuple(Box(low=lo, high=hi, shape=(D,), dtype=np.float32) for _ in range(N)))
    return actions, observations
