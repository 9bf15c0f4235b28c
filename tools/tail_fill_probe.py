#!/usr/bin/env python3
"""can the ragged end of the chip-filling launch be filled?  The thread-per-env launch at 524 288 envs is four
generations of waves whose durations spread from 50 k to 126 k ticks: 165 us where 117 us of work exists.
The two step kernels are bit-identical, so a batch may be split: the first (1 - f) E envs on the thread-per-env kernel (a
high-priority stream), the last f E on the lane-group kernel (a lower-priority stream), whose short waves the dispatcher can
place into the wave slots the big grid leaves idle while its last waves run; one event join per step.

    python tools/tail_fill_probe.py [--envs 524288] [--fractions 0 0.05 0.1 0.15 0.2 0.3] [--steps 60]

Prints one JSON line per split (profiles/r4_tail_fill.jsonl).  f = 0 is the unsplit launch through the same loop."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from marbler_amd import VecRobotariumEnv  # noqa: E402

OV = {"predator": 3, "capture": 2, "n_agents": 5}


def make(kernel, E, offset):
    os.environ["RG_STEP_KERNEL"] = kernel
    env = VecRobotariumEnv("PredatorCapturePrey", E, overrides=OV, seed=0, env_offset=offset)
    assert env.step_kernel == kernel
    return env


def run(E, f, steps, warm, same_priority=False, group_first=False):
    dev = torch.device("cuda:0")
    E2 = int(round(E * f / 64)) * 64
    E1 = E - E2
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    s_tpe = torch.cuda.Stream(dev, priority=0 if same_priority else hi)
    s_grp = torch.cuda.Stream(dev, priority=0 if same_priority else lo)
    acts = torch.randint(0, 5, (8, E, 5), device=dev, dtype=torch.int32)
    with torch.cuda.stream(s_tpe):
        a = make("tpe", E1, 0)
        a.reset()
    b = None
    if E2:
        with torch.cuda.stream(s_grp):
            b = make("group", E2, E1)
            b.reset()
    torch.cuda.synchronize()
    main = torch.cuda.current_stream(dev)
    ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()

    def step(i):
        x = acts[i % 8]
        s_tpe.wait_stream(main)
        if b is not None:
            s_grp.wait_stream(main)
        order = ((b, s_grp, x[E1:], ev_b), (a, s_tpe, x[:E1], ev_a)) if group_first else ((a, s_tpe, x[:E1], ev_a), (b, s_grp, x[E1:], ev_b))
        for env, st, xa, ev in order:
            if env is None:
                continue
            with torch.cuda.stream(st):
                env.step(xa)
                ev.record(st)
        main.wait_event(ev_a)
        if b is not None:
            main.wait_event(ev_b)

    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(main)
    for i in range(steps):
        step(i)
    t1.record(main)
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 1e3 / steps
    done = int(a.done_count.sum()) + (int(b.done_count.sum()) if b is not None else 0)
    out = {"envs": E, "f_group": f, "envs_tpe": E1, "envs_group": E2, "us_per_step": round(us, 2), "g_agent_steps_per_s": round(E * 5 / us * 1e-3, 3),
           "hbm_frac_algorithmic": round(585 * E / (us * 1e-6) / 8e12, 4), "episodes_finished": done, "same_priority": same_priority, "group_first": group_first,
           "priority_range": [lo, hi]}
    a.close()
    if b is not None:
        b.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=524288)
    ap.add_argument("--fractions", type=float, nargs="*", default=[0.0, 0.05, 0.1, 0.15, 0.2, 0.3])
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=30)
    args = ap.parse_args()
    for f in args.fractions:
        for kw in ({},) if f == 0 else ({}, {"same_priority": True}, {"group_first": True}):
            print(json.dumps(run(args.envs, f, args.steps, args.warmup, **kw)), flush=True)
