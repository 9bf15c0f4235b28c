"""Initial conditions drawn exactly as the reference draws them (parity mode of reset, row a17).

The reference seeds NumPy's GLOBAL legacy generator once, in the scenario constructor
(`np.random.seed(args.seed)`: PredatorCapturePrey.py:27-28, warehouse.py:59-60,
MaterialTransport.py:65-66, simple.py, ArcticTransport.py) and every `reset()` then consumes that
stream through `utilities/misc.py:49-63 generate_initial_locations` -> rps
`generate_initial_conditions` (`np.random.choice(cells, N, replace=False)`,
then one `np.random.rand()` per robot for the heading) and, for MaterialTransport, two
`np.random.normal` calls for the zone loads (MaterialTransport.py:99-100).

The device sampler (csrc/device_common.h reset_group) keeps one Philox stream per (env, episode) --
the throughput path, invariant to sharding -- and therefore starts episodes elsewhere than the
reference does for the same `seed`.  This module is the parity path: the same calls, in the same
order, on a `np.random.RandomState(seed)` (bit-identical to the seeded global generator), on the
host; `VecRobotariumEnv.reset(reference_rng=...)` uploads the result.  `Wrapper` uses it whenever
the config carries `seed != -1`, so `Wrapper(seed=s)` starts every episode where the reference's
`Wrapper(seed=s)` does GIVEN THE RESTATED rps sampler: the reference's own call sequence
(misc.py:49-63, the scenario resets) is pinned by the first rows of every episode of tests/golden/*.npz,
but those vectors were recorded over oracle/rps_restated, so `generate_initial_conditions` itself
(below) is checked only against the restatement -- parity with real rps is UNPINNED, like rows a4-a10.
It follows upstream rps as recalled, including the `+ 1` on the sampled cell indices
(`choices = np.random.choice(x_range * y_range, N, replace=False) + 1`: cell (0, 0) is never used, the
first cell of column x_range is); rounds 1-2 of this repo omitted that shift.
"""
import numpy as np


def generate_initial_conditions(rng, N, spacing=0.3, width=3, height=1.8):
    """rps.utilities.misc.generate_initial_conditions (Appendix A.7), float64, drawing from `rng`."""
    x_range = int(np.floor(width / spacing))
    y_range = int(np.floor(height / spacing))
    assert x_range != 0 and y_range != 0, "spacing too large for the area"
    assert x_range * y_range > N, "more robots than grid cells"
    choices = (rng.choice(x_range * y_range, N, replace=False) + 1)   # upstream's "+ 1": indices 1 .. cells
    poses = np.zeros((3, N))
    for i, c in enumerate(choices):
        x, y = divmod(c, y_range)
        poses[0, i] = x * spacing - width / 2
        poses[1, i] = y * spacing - height / 2
        poses[2, i] = rng.rand() * 2 * np.pi - np.pi
    return poses


def generate_initial_locations(rng, num_locs, width, height, thresh, start_dist=.3, spawn_left=True):
    """utilities/misc.py:49-63."""
    poses = generate_initial_conditions(rng, num_locs, spacing=start_dist, width=width, height=height)
    for i in range(poses.shape[1]):
        if spawn_left:
            poses[0][i] -= (width / 2 - thresh)
        else:
            poses[0][i] += (width / 2 - thresh)
        poses[2][i] = 0
    return poses


def draw_reset(scenario, cfg, rng, py_random=None):
    """One scenario.reset() worth of draws.  Returns a dict of float64 / int arrays:
    poses [3,N]; PredatorCapturePrey: prey_loc [P,2]; Simple: prey_loc [1,2] (its goal);
    MaterialTransport: zone_load [2]; ArcticTransport: grid [96] uint8, goal_col.
    `py_random`: a `random.Random` for ArcticTransport's goal column (ArcticTransport.py:72 draws it from
    Python's `random`); None = the `random` module itself, like the reference."""
    L, R, U, D = cfg["LEFT"], cfg["RIGHT"], cfg["UP"], cfg["DOWN"]
    out = {}
    if scenario == "PredatorCapturePrey":       # PredatorCapturePrey.py:114-136
        N = int(cfg["predator"]) + int(cfg["capture"])
        width = cfg["ROBOT_INIT_RIGHT_THRESH"] - L
        height = D - U
        out["poses"] = generate_initial_locations(rng, N, width, height, cfg["ROBOT_INIT_RIGHT_THRESH"],
                                                  start_dist=cfg["start_dist"])
        width = R - cfg["PREY_INIT_LEFT_THRESH"]
        prey = generate_initial_locations(rng, int(cfg["num_prey"]), width, height, cfg["ROBOT_INIT_RIGHT_THRESH"],
                                          start_dist=cfg["step_dist"], spawn_left=False)
        out["prey_loc"] = prey[:2].T.copy()
    elif scenario == "Warehouse":               # warehouse.py:84-100
        width, height = R - L, D - U
        poses = generate_initial_conditions(rng, int(cfg["n_agents"]), spacing=cfg["start_dist"], width=width, height=height)
        poses[0] += (1.5 + L) / 2
        poses[0] -= (1.5 - R) / 2
        poses[1] -= (1 + U) / 2
        poses[1] += (1 - D) / 2
        out["poses"] = poses
    elif scenario == "MaterialTransport":       # MaterialTransport.py:94-111
        z = []
        for key in ("zone1", "zone2"):
            args = dict(cfg[key])
            dist = args.pop("distribution")
            z.append(int(getattr(rng, dist)(**args)))
        out["zone_load"] = np.array(z, dtype=np.int64)
        width, height = cfg["end_goal_width"], D - U
        out["poses"] = generate_initial_locations(rng, int(cfg["n_agents"]), width, height, L + cfg["end_goal_width"],
                                                  start_dist=cfg["start_dist"])
    elif scenario == "Simple":                  # simple.py:129-153
        width = cfg["ROBOT_INIT_RIGHT_THRESH"] - L
        height = D - U
        out["poses"] = generate_initial_locations(rng, int(cfg["n_agents"]), width, height,
                                                  cfg["ROBOT_INIT_RIGHT_THRESH"], start_dist=cfg["start_dist"])
        width = R - cfg["PREY_INIT_LEFT_THRESH"]
        goal = generate_initial_locations(rng, 1, width, height, cfg["ROBOT_INIT_RIGHT_THRESH"],
                                          start_dist=cfg["step_dist"], spawn_left=False)
        out["prey_loc"] = goal[:2].T.copy()
    elif scenario == "ArcticTransport":         # ArcticTransport.py:56-82 (start poses :30-33)
        import random as _random
        pr = py_random if py_random is not None else _random
        out["poses"] = np.array([[-0.3, 0.3, -0.9, 0.9], [-0.8] * 4, [np.pi / 2] * 4])
        grid = rng.randint(3, size=(8, 12))
        gc = pr.randint(1, 11)
        grid[0][gc] = grid[0][gc - 1] = grid[1][gc] = grid[1][gc - 1] = 3
        grid[7][1:11] = 0
        out["grid"] = grid.reshape(-1).astype(np.uint8)
        out["goal_col"] = int(gc)
    else:
        raise KeyError(scenario)
    # roboEnv._create_robotarium ends in `self.robotarium.step()` at zero velocity (utilities/roboEnv.py:109-111): positions stay,
    # the headings pass once through rps' wrap theta = arctan2(sin theta, cos theta) -- the identity up to an ulp, and not always
    # to the last bit (a Warehouse start with seed 261 moves by 7e-18)
    th = out["poses"][2]
    out["poses"][2] = np.arctan2(np.sin(th), np.cos(th))
    return out
