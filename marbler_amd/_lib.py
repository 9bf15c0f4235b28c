"""ctypes binding of include/robogym.h (librobogym_hip.so).

This is the "FFI stub" a maintainer of the reference would add: the
reference is pure Python, so the boundary is Python -> C ABI.  There is NO CPU fallback:
if the library is missing, or no HIP device is visible, construction fails loudly.
"""
import ctypes as C
import os

MAX_AGENTS, MAX_PREY = 16, 64
ABI_VERSION = 6
RESET_BOOK_EPISODE = 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ROBOGYM_LIB") or os.path.join(_HERE, "librobogym_hip.so")  # override: diagnostic builds


class RgGrid(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("spacing", C.c_float), ("w2", C.c_float),
                ("h2", C.c_float), ("ox1", C.c_float), ("ox2", C.c_float), ("oy1", C.c_float),
                ("oy2", C.c_float)]


class RgScenarioParams(C.Structure):
    _fields_ = [
        ("scenario", C.c_int32), ("n_agents", C.c_int32), ("obs_dim", C.c_int32),
        ("update_frequency", C.c_int32), ("controller_period", C.c_int32),
        ("max_episode_steps", C.c_int32), ("penalize_violations", C.c_int32),
        ("barrier_has_unsafe_gain", C.c_int32), ("collision_variant", C.c_int32),
        ("capability_aware", C.c_int32), ("num_prey", C.c_int32), ("num_neighbors", C.c_int32),
        ("torque", C.c_int32 * MAX_AGENTS),
        ("time_step", C.c_float), ("bound_x0", C.c_float), ("bound_y0", C.c_float),
        ("bound_w", C.c_float), ("bound_h", C.c_float),
        ("robot_diameter", C.c_float), ("wheel_radius", C.c_float), ("max_linear_velocity", C.c_float),
        ("collision_offset", C.c_float), ("collision_diameter", C.c_float),
        ("projection_distance", C.c_float), ("angular_velocity_limit", C.c_float),
        ("position_velocity_limit", C.c_float),
        ("barrier_gain", C.c_float), ("unsafe_barrier_gain", C.c_float), ("safety_radius", C.c_float),
        ("barrier_magnitude_limit", C.c_float), ("qp_rtol", C.c_float), ("qp_max_sweeps", C.c_int32),
        ("left", C.c_float), ("right", C.c_float), ("up", C.c_float), ("down", C.c_float),
        ("agent_step", C.c_float * MAX_AGENTS), ("sensing_radius", C.c_float * MAX_AGENTS),
        ("capture_radius", C.c_float * MAX_AGENTS),
        ("time_penalty", C.c_float), ("sense_reward", C.c_float), ("capture_reward", C.c_float),
        ("violation_reward", C.c_float),
        ("load_reward", C.c_float), ("unload_reward", C.c_float), ("goal_width", C.c_float),
        ("unload_multiplier", C.c_float), ("load_multiplier", C.c_float), ("end_goal_width", C.c_float),
        ("zone1_radius", C.c_float), ("reward_scaler", C.c_float),
        ("arctic_normal_step", C.c_float), ("arctic_slow_step", C.c_float), ("arctic_fast_step", C.c_float),
        ("not_reached_penalty", C.c_float), ("dist_multiplier", C.c_float),
        ("agent_grid", RgGrid), ("prey_grid", RgGrid), ("keep_theta", C.c_int32), ("shared_reward", C.c_int32),
        ("zone1_mean", C.c_float), ("zone1_std", C.c_float), ("zone2_mean", C.c_float),
        ("zone2_std", C.c_float),
        ("qp_mode", C.c_int32), ("ipm_abstol", C.c_float), ("ipm_reltol", C.c_float), ("ipm_feastol", C.c_float), ("ipm_maxiters", C.c_int32),
    ]


class RgState(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("carry_dist", C.c_void_p), ("episode_steps", C.c_void_p),
                ("reset_count", C.c_void_p), ("prey_loc", C.c_void_p), ("prey_sensed", C.c_void_p),
                ("prey_captured", C.c_void_p), ("loaded", C.c_void_p), ("load", C.c_void_p),
                ("zone_load", C.c_void_p), ("messages", C.c_void_p), ("grid", C.c_void_p),
                ("goal_col", C.c_void_p), ("pixel_type", C.c_void_p), ("reached_goal", C.c_void_p),
                ("ep_return", C.c_void_p),
                ("done_return_sum", C.c_void_p), ("done_count", C.c_void_p), ("done_steps_sum", C.c_void_p),
                ("next_init", C.c_void_p), ("next_episode", C.c_void_p)]


class RgStepIO(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p),
                ("dist_travelled", C.c_void_p), ("violation", C.c_void_p), ("remaining", C.c_void_p),
                ("qp_sweeps", C.c_void_p),
                ("elapsed", C.c_void_p), ("truncated", C.c_void_p), ("ended", C.c_void_p), ("reward_sum", C.c_void_p),
                ("time_limit", C.c_int32), ("zero_obs_on_end", C.c_int32)]


class RgActorWeights(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("wih", C.c_void_p), ("bih", C.c_void_p),
                ("whh", C.c_void_p), ("bhh", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("n_sets", C.c_int32), ("input_dim", C.c_int32), ("hidden_dim", C.c_int32),
                ("n_actions", C.c_int32), ("use_rnn", C.c_int32), ("gru_packed", C.c_int32)]


EXPORTS = ("rg_abi_version", "rg_last_error", "rg_sizeof_params", "rg_sizeof_state", "rg_sizeof_step_io", "rg_next_init_stride",
           "rg_create", "rg_destroy", "rg_bind_state", "rg_set_stream", "rg_reset", "rg_step", "rg_rollout", "rg_get_obs", "rg_step_kernel",
           "rg_actor_forward", "rg_actor_forward_explore", "rg_actor_pack_gru", "rg_actor_pack_gru_bf16x3", "rg_actor_pack_gru_f16x2", "rg_actor_last_error")

_lib = None


class RobogymError(RuntimeError):
    pass


def load():
    """Load librobogym_hip.so and type its entry points.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RobogymError(
            f"{LIB_PATH} not found: build it with `python -m marbler_amd.build` (hipcc, gfx950). "
            "marbler_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RobogymError(f"{LIB_PATH} does not export {name}")
    lib.rg_abi_version.restype = C.c_int
    lib.rg_last_error.restype = C.c_char_p
    lib.rg_create.restype = C.c_void_p
    lib.rg_create.argtypes = [C.POINTER(RgScenarioParams), C.c_int32, C.c_int64, C.c_int32, C.c_void_p]
    lib.rg_next_init_stride.argtypes = [C.POINTER(RgScenarioParams)]
    lib.rg_next_init_stride.restype = C.c_int
    lib.rg_destroy.argtypes = [C.c_void_p]
    lib.rg_bind_state.argtypes = [C.c_void_p, C.POINTER(RgState)]
    lib.rg_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.rg_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32]
    lib.rg_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RgStepIO), C.c_int32, C.c_uint64]
    lib.rg_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(RgStepIO), C.c_int32, C.c_uint64]
    lib.rg_get_obs.argtypes = [C.c_void_p, C.c_void_p]
    lib.rg_step_kernel.argtypes = [C.c_void_p]
    lib.rg_step_kernel.restype = C.c_int
    lib.rg_actor_forward.argtypes = [C.POINTER(RgActorWeights), C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rg_actor_forward.restype = C.c_int
    lib.rg_actor_forward_explore.argtypes = [C.POINTER(RgActorWeights), C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    lib.rg_actor_forward_explore.restype = C.c_int
    lib.rg_actor_pack_gru.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rg_actor_pack_gru.restype = C.c_int
    lib.rg_actor_pack_gru_bf16x3.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rg_actor_pack_gru_bf16x3.restype = C.c_int
    lib.rg_actor_pack_gru_f16x2.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rg_actor_pack_gru_f16x2.restype = C.c_int
    lib.rg_actor_last_error.restype = C.c_char_p
    for f in (lib.rg_destroy, lib.rg_bind_state, lib.rg_set_stream, lib.rg_reset, lib.rg_step, lib.rg_rollout, lib.rg_get_obs,
              lib.rg_sizeof_params, lib.rg_sizeof_state, lib.rg_sizeof_step_io):
        f.restype = C.c_int
    if lib.rg_abi_version() != ABI_VERSION:
        raise RobogymError(f"ABI version {lib.rg_abi_version()} != {ABI_VERSION}; rebuild the library")
    if (lib.rg_sizeof_params() != C.sizeof(RgScenarioParams) or lib.rg_sizeof_state() != C.sizeof(RgState)
            or lib.rg_sizeof_step_io() != C.sizeof(RgStepIO)):
        raise RobogymError("struct layout of the binding differs from the compiled library; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RobogymError(f"{what} failed ({rc}): {load().rg_last_error().decode()}")
