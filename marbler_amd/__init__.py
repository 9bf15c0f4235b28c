"""marbler_amd -- MI355X-native (gfx950) vectorised Robotarium-gym step engine.

The one hot path of GT-STAR-Lab/MARBLER (the multi-robot env step) as hand-written HIP
kernels behind the reference's reset()/step() surface.  See DESIGN.md and INTEGRATION.md.
There is no CPU fallback: the HIP library must be built (python -m marbler_amd.build) and a
GPU must be visible, or construction raises.
"""
from ._lib import RobogymError  # noqa: F401
from .params import load_config, make_params  # noqa: F401
from .vec_env import VecRobotariumEnv  # noqa: F401
from .wrapper import Wrapper, env_dict, register_gym_ids  # noqa: F401

__all__ = ["VecRobotariumEnv", "Wrapper", "env_dict", "register_gym_ids", "load_config", "make_params",
           "RobogymError"]
