#!/usr/bin/env python3
"""Can the evaluation loop overlap one half's policy step (MFMA-bound) with the other half's env step (one dependent chain per
wave)?  E envs as ONE handle on one stream (actor -> step -> actor -> ...) against TWO handles of E / 2 (env_offset keeps every
env's random streams: the halves are the same envs) on two streams, each running its own actor -> step chain.
    python tools/eval_pipeline_probe.py [--envs 4096 8192] [--hidden 128] [--steps 300]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from marbler_amd import VecRobotariumEnv  # noqa: E402
from marbler_amd.evaluate import BatchedActor  # noqa: E402
from test_gpu_actor import _random_actor  # noqa: E402

OV = {"predator": 2, "capture": 2, "n_agents": 4}


def run(E, parts, H, steps, warm=50):
    dev = torch.device("cuda:0")
    Ep = E // parts
    halves = []
    for p in range(parts):
        st = torch.cuda.Stream(dev) if parts > 1 else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            env = VecRobotariumEnv("PredatorCapturePrey", Ep, overrides=OV, seed=0, env_offset=p * Ep)
            env.set_stream(st)
            env.reset()
            actor = BatchedActor(_random_actor(1, env.D + env.N, H, 5, True, 3), env.N, device=dev)
            halves.append({"env": env, "actor": actor, "st": st, "hidden": actor.init_hidden(Ep),
                           "q": torch.empty(Ep, env.N, 5, device=dev), "act": torch.zeros(Ep, env.N, dtype=torch.int32, device=dev)})
    for h in halves:
        h["ptr"] = h["act"].data_ptr()
    torch.cuda.synchronize()

    def iteration():
        for h in halves:
            env = h["env"]
            h["actor"].forward_fused(env.obs, h["hidden"], restart=env.done_u8, q_out=h["q"], actions_out=h["act"], stream=h["st"])
            env.step_raw(h["ptr"])

    for _ in range(warm):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    done = sum(int(h["env"].done_count.sum()) for h in halves)
    ret = sum(float(h["env"].done_return_sum.sum()) for h in halves)
    for h in halves:
        h["env"].close()
    return {"envs": E, "handles": parts, "hidden": H, "us_per_iteration": round(t / steps * 1e6, 2), "host_us_per_iteration": round(t_host / steps * 1e6, 2),
            "M_agent_steps_per_s": round(E * 4 / (t / steps) * 1e-6, 1), "episodes": done, "return_sum": ret}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="*", default=[4096, 8192])
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--steps", type=int, default=300)
    # (one configuration per process: with more HIP streams alive than hardware queues, later cases in the same process run
    # several times slower -- an artefact of the probe, not of the loop)
    ap.add_argument("--handles", type=int, default=0)
    a = ap.parse_args()
    if a.handles:
        for E in a.envs:
            print(json.dumps(run(E, a.handles, a.hidden, a.steps)), flush=True)
    else:
        import subprocess
        for E in a.envs:
            for parts in (1, 2):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--envs", str(E), "--handles", str(parts), "--hidden", str(a.hidden), "--steps", str(a.steps)])
