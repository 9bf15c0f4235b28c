"""register() records its calls (TEST STAND-IN, see gym/__init__.py)."""
REGISTRY = {}
CALLS = []


class EnvSpec(object):
    def __init__(self, id, entry_point=None, kwargs=None, **rest):
        self.id = id
        self.entry_point = entry_point
        self.kwargs = dict(kwargs or {})
        self.max_episode_steps = rest.get("max_episode_steps")


def register(id, **kwargs):
    if id in REGISTRY:
        raise ValueError(f"Cannot re-register id: {id}")
    CALLS.append((id, dict(kwargs)))
    REGISTRY[id] = EnvSpec(id, **kwargs)
