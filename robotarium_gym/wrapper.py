"""`robotarium_gym.wrapper:Wrapper` -- the entry point string the reference registers."""
from marbler_amd.wrapper import Wrapper, env_dict  # noqa: F401
