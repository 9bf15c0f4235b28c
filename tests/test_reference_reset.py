"""Row a17, parity mode: marbler_amd.reference_reset draws initial conditions from a seeded legacy NumPy
stream with the reference's own call sequence (misc.py:49-63 + rps generate_initial_conditions, scenario
reset()s).  What the reference itself pins is its CALL SEQUENCE: the free-running golden vectors were recorded
from `Wrapper(seed=s)`, so the pre-step state of the FIRST step of every episode is what the reference's
reset() produced over the restated rps -- all episodes of all seeds of every fixture must be reproduced,
float64-equal.  rps' `generate_initial_conditions` itself is the builder's restatement on both sides of this
comparison (oracle/rps_restated, marbler_amd/reference_reset.py): parity with real rps is unpinned for it."""
import os
import random

import numpy as np
import pytest

from helpers import golden_files, load_golden

FREE_RUNNING = [p for p in golden_files() if "viol_" not in os.path.basename(p)]


@pytest.mark.parametrize("path", FREE_RUNNING, ids=lambda p: os.path.basename(p)[:-4])
def test_reset_draws_equal_the_reference_episode_starts(path):
    from marbler_amd.reference_reset import draw_reset
    g, scenario, cfg = load_golden(path)
    per = int(g["steps_per_seed"])
    first = g["first_after_reset"].astype(bool)
    episodes = 0
    for si, seed in enumerate(g["seeds"]):
        rng = np.random.RandomState(int(seed))              # = np.random.seed(args.seed) in the scenario constructor
        pyr = random.Random(int(seed) + 12345)              # ref_harness seeds Python's `random` for ArcticTransport
        for t in range(si * per, (si + 1) * per):
            if not first[t]:
                continue
            d = draw_reset(scenario, dict(cfg, seed=int(seed)), rng, py_random=pyr)
            assert np.array_equal(d["poses"], g["pre_poses"][t]), (int(seed), t)
            if "prey_loc" in d:
                assert np.array_equal(d["prey_loc"], g["pre_prey_loc"][t]), (int(seed), t)
            if "zone_load" in d:
                assert np.array_equal(d["zone_load"], g["pre_zone_load"][t]), (int(seed), t)
            if "grid" in d:
                assert np.array_equal(d["grid"], g["pre_grid"][t]) and d["goal_col"] == int(g["pre_goal_col"][t])
            episodes += 1
    assert episodes >= len(g["seeds"])
    if scenario in ("PredatorCapturePrey", "Simple", "ArcticTransport"):
        assert episodes > len(g["seeds"]), "the fixture should hold more than one episode per seed"
