"""Free-running statistical equivalence, CPU tier: the float32 oracle (what the kernels compute, bit for bit) against the
float64 oracle (the reference's arithmetic) from one reset stream and one action stream, at a size this tier can afford;
and the committed record tests/golden/FREE_RUNNING_STATS.json (full size) is self-consistent.  The `-m gpu` tier
(tests/test_gpu_free_running_stats.py) runs the kernels themselves at full size against that record."""
import json
import os

import numpy as np
import pytest

import free_running as fr

RECORD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "FREE_RUNNING_STATS.json")


@pytest.mark.parametrize("name", sorted(fr.CASES))
def test_float32_and_float64_oracles_agree_statistically(name, oracle_lib):
    from marbler_amd.params import load_config, make_params
    scenario, ov, n_act, E, steps = fr.CASES[name]
    E, steps = 768, 260
    cfg = load_config(scenario, None, ov)
    p = make_params(scenario, cfg)
    f64 = fr.run_oracle(oracle_lib, scenario, cfg, p, E, steps, n_act, np.float64, fr.SEED, fr.ACTION_SEED, threads=4)
    f32 = fr.run_oracle(oracle_lib, scenario, cfg, p, E, steps, n_act, np.float32, fr.SEED, fr.ACTION_SEED, threads=4)
    fr.compare(f32, f64, name)
    assert f32["episodes"] > 1000 and sum(f32["violation_counts"][1:]) > 20


def test_committed_record_is_self_consistent():
    rec = json.load(open(RECORD))
    assert set(rec["cases"]) == set(fr.CASES) and rec["seed"] == fr.SEED and rec["action_seed"] == fr.ACTION_SEED
    for name, c in rec["cases"].items():
        assert (c["envs"], c["steps"]) == fr.CASES[name][3:]
        fr.compare(c["float32"], c["float64"], name)
        assert c["float64"]["env_steps"] == c["envs"] * c["steps"]
