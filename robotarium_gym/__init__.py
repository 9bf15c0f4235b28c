"""Import-name shim: EPyMARL resolves `env_args.key="robotarium_gym:PredatorCapturePrey-v0"` by
importing `robotarium_gym` (the reference's package name, robotarium_gym/__init__.py:4-23) and
expecting the ids to be registered as a side effect.  With this repo on PYTHONPATH instead of
the reference, the same key constructs the HIP-backed Wrapper."""
from marbler_amd.wrapper import register_gym_ids

REGISTERED = register_gym_ids(entry_point="robotarium_gym.wrapper:Wrapper")
