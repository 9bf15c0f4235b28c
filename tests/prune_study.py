"""Study (CPU, C oracle): how many barrier rows can be pruned before the Hildreth sweeps, per env and per lane-group wave.
python tests/prune_study.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import c_oracle
from helpers import oracle_reset, oracle_reset_params
from marbler_amd.params import load_config, make_params

c_oracle.build_library()
scenario, ov, n_act = "PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5
cfg = load_config(scenario, None, ov)
p = make_params(scenario, cfg)
E, T = 4096, 120
env = c_oracle.OracleVecEnv(scenario, cfg, E, dtype=np.float32)
rp = oracle_reset_params(c_oracle, p)
for e in range(E):
    oracle_reset(c_oracle, env, rp, 0, e, 0)
episodes = np.zeros(E, np.int64)
rng = np.random.RandomState(1)
N = env.N
iu = np.triu_indices(N, 1)
r2, pd = 0.2 ** 2, 0.05
bound = np.sqrt(2 * N) * 0.15 * 1.001
rows = []
for t in range(T):
    P = env.poses.copy()                                   # [E, 3, N]
    xi = P[:, :2] + pd * np.stack([np.cos(P[:, 2]), np.sin(P[:, 2])], 1)
    d = xi[:, :, :, None] - xi[:, :, None, :]
    ee = (d ** 2).sum(1)[:, iu[0], iu[1]]                  # [E, pairs]
    h = ee - r2
    beta = 50.0 * h ** 3
    keep = ~(beta > np.sqrt(ee) * bound)
    unsafe = (h < 0).any(1)
    keep_env = np.where(unsafe[:, None], True, keep)
    act = rng.randint(0, n_act, size=(E, N)).astype(np.int32)
    env.step(act, threads=8)
    rows.append((keep.sum(1), unsafe, env.qp_sweeps.copy(), keep_env))
    for e in np.nonzero(env.done)[0]:
        episodes[e] += 1
        oracle_reset(c_oracle, env, rp, 0, e, int(episodes[e]))
kept = np.concatenate([r[0] for r in rows]); unsafe = np.concatenate([r[1] for r in rows]); sw = np.concatenate([r[2] for r in rows])
keep_env = np.concatenate([r[3] for r in rows])
print("pairs kept per env (of 10): mean %.2f; hist" % kept.mean(), np.bincount(kept, minlength=11))
print("unsafe envs: %.4f" % unsafe.mean())
for lo, hi in ((1, 2), (3, 4), (5, 8), (9, 12), (13, 99)):
    m = (sw >= lo) & (sw <= hi)
    if m.any():
        print(f"sweeps {lo}-{hi}: {m.mean():.4f} of env steps; unsafe {unsafe[m].mean():.3f}; kept pairs mean {kept[m].mean():.2f} (safe only: {kept[m & ~unsafe].mean() if (m & ~unsafe).any() else float('nan'):.2f})")
# lane-group rounds: XOR round k holds pairs (a, a^k); a round is needed by a wave when any of its envs keeps a pair of it
pair_round = np.array([i ^ j for i, j in zip(*iu)])
for epw in (4, 8):
    W = keep_env.shape[0] // epw
    ke = keep_env[:W * epw].reshape(W, epw, -1)
    need = np.stack([(ke[:, :, pair_round == k]).any(axis=(1, 2)) for k in range(1, 8)], 1)
    print(f"{epw} envs per wave: rounds needed mean {need.sum(1).mean():.2f} of 7; by round", need.mean(0).round(2))
    sww = sw[:W * epw].reshape(W, epw).max(1)
    for lo, hi in ((1, 2), (3, 6), (7, 99)):
        m = (sww >= lo) & (sww <= hi)
        if m.any():
            print(f"   waves with max sweeps {lo}-{hi}: {m.mean():.3f}; rounds needed {need[m].sum(1).mean():.2f}")
