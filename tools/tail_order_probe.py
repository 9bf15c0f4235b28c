#!/usr/bin/env python3
"""Would a better DISPATCH ORDER shorten the chip-filling launch's ragged end?  Per-wave durations of the thread-per-env kernel
at 524 288 envs (-DRG_STAMPS build: s_memtime ticks, written over qp_sweeps) for consecutive steps, then a list-scheduling
simulation (2048 wave slots, workgroups started in order as slots free up -- what the dispatcher does) of
  * the order as launched (chunk index),
  * longest-predicted-first, the prediction being the same wave's duration k steps earlier (k = 1, 4, 16),
  * the oracle order (this step's own durations: the bound).
    RG_STEP_KERNEL=tpe python tools/tail_order_probe.py [--envs 524288] [--steps 48]"""
import argparse, heapq, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ROBOGYM_LIB"] = os.environ.get("RG_STAMPS_LIB") or os.path.join(ROOT, "marbler_amd", "librobogym_stamps.so")
os.environ["RG_STEP_KERNEL"] = "tpe"
sys.path.insert(0, ROOT)
import numpy as np
import torch
from marbler_amd import VecRobotariumEnv

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=524288)
ap.add_argument("--steps", type=int, default=48)
ap.add_argument("--slots", type=int, default=2048)
args = ap.parse_args()
E = args.envs
env = VecRobotariumEnv("PredatorCapturePrey", E, overrides={"predator": 3, "capture": 2, "n_agents": 5}, collect_qp_stats=True)
acts = torch.randint(0, 5, (8, E, 5), device=env.device, dtype=torch.int32)
env.reset()
for i in range(60):
    env.step(acts[i % 8])
dur = []
for i in range(args.steps):
    env.step(acts[i % 8])
    dur.append(env.qp_sweeps.view(-1, 64)[:, 5].cpu().numpy().astype(np.int64))     # slot 5: the wave's step is stored
dur = np.stack(dur)                                                                   # [steps, waves]
np.save(os.path.join(ROOT, "gpurun_out", "r4_tpe_wave_ticks.npy"), dur.astype(np.int32))


def makespan(d, order, slots):
    free = [0] * slots
    heapq.heapify(free)
    end = 0
    for w in order:
        t = heapq.heappop(free) + int(d[w])
        end = max(end, t)
        heapq.heappush(free, t)
    return end


W = dur.shape[1]
out = {"envs": E, "waves": W, "slots": args.slots, "steps": args.steps, "ticks_mean": float(dur.mean()), "ticks_p1": float(np.percentile(dur, 1)),
       "ticks_p99": float(np.percentile(dur, 99)), "ideal_ticks": float(dur.sum(axis=1).mean() / args.slots)}
res = {"as_launched": [], "oracle_lpt": []}
for k in (1, 4, 16):
    res[f"lpt_by_{k}_steps_ago"] = []
for t in range(16, args.steps):
    d = dur[t]
    res["as_launched"].append(makespan(d, range(W), args.slots))
    res["oracle_lpt"].append(makespan(d, np.argsort(-d), args.slots))
    for k in (1, 4, 16):
        res[f"lpt_by_{k}_steps_ago"].append(makespan(d, np.argsort(-dur[t - k]), args.slots))
out["makespan_ticks_mean"] = {k: float(np.mean(v)) for k, v in res.items()}
out["corr_step_to_step"] = {str(k): float(np.mean([np.corrcoef(dur[t], dur[t - k])[0, 1] for t in range(16, args.steps)])) for k in (1, 4, 16)}
print(json.dumps(out))
