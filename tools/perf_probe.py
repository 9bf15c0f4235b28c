#!/usr/bin/env python3
"""Kernel-time probe: per-launch time of rg_step under variations (sweep cap, auto-reset, E).
Diagnostic only (not the judged bench)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from marbler_amd import VecRobotariumEnv  # noqa: E402

OV = {"PredatorCapturePrey": {"predator": 3, "capture": 2, "n_agents": 5},
      "Warehouse": {"n_agents": 8},
      "MaterialTransport": {"n_agents": 6, "n_fast_agents": 3, "n_slow_agents": 3, "start_dist": 0.25}}


def probe(scenario, E, steps=400, warm=100, auto_reset=True, **over):
    ov = dict(OV[scenario])
    ov.update(over)
    env = VecRobotariumEnv(scenario, E, overrides=ov, auto_reset=auto_reset, collect_qp_stats=True)
    nact = 20 if scenario == "MaterialTransport" else 5
    g = torch.Generator(device=env.device)
    g.manual_seed(1)
    acts = torch.randint(0, nact, (64, E, env.N), generator=g, device=env.device, dtype=torch.int32)
    env.reset()
    for i in range(warm):
        env.step_raw(acts[i % 64].data_ptr())
        if not auto_reset and i % 20 == 19:
            env.reset(env.done_u8)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sw = torch.zeros(64, dtype=torch.int64, device=env.device)
    a.record()
    for i in range(steps):
        env.step_raw(acts[i % 64].data_ptr())
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    # sweep statistics of the last step
    q = env.qp_sweeps.float()
    return {"scenario": scenario, "E": E, "auto_reset": auto_reset, "over": over, "us_per_step": ms * 1e3,
            "agent_steps_per_s": E * env.N / (ms * 1e-3), "qp_sweeps_max": int(q.max().item()),
            "qp_sweeps_mean": float(q.mean().item())}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="basic")
    args = ap.parse_args()
    out = []
    if args.set == "basic":
        for cap in (40, 16, 8, 4, 2, 1):
            out.append(probe("PredatorCapturePrey", 4096, qp_max_sweeps=cap))
        out.append(probe("PredatorCapturePrey", 4096, auto_reset=False))
        for E in (8, 512, 2048, 8192, 32768, 131072, 524288):
            out.append(probe("PredatorCapturePrey", E, steps=200 if E < 100000 else 50))
        out.append(probe("Warehouse", 4096))
        out.append(probe("MaterialTransport", 4096, steps=200))
    if args.set == "scale":   # both step kernels across batch sizes (RG_STEP_KERNEL is read at rg_create)
        for kern in ("group", "tpe"):
            os.environ["RG_STEP_KERNEL"] = kern
            for scn, Es in (("PredatorCapturePrey", (256, 1024, 2048, 4096, 8192, 16384, 65536, 131072, 524288)),
                            ("Warehouse", (4096, 65536)), ("MaterialTransport", (4096, 65536))):
                for E in Es:
                    r = probe(scn, E, steps=200 if E < 100000 else 50)
                    r["kernel"] = kern
                    out.append(r)
                    print(json.dumps(r), flush=True)
        out = []
    if args.set == "cross":   # cross-over between the two step kernels per agent count
        for kern in ("group", "tpe"):
            os.environ["RG_STEP_KERNEL"] = kern
            cases = [("PredatorCapturePrey", E, {}) for E in (24576, 32768, 49152)]
            cases += [("PredatorCapturePrey", E, {"predator": 2, "capture": 2, "n_agents": 4}) for E in (32768, 131072, 524288)]
            cases += [("PredatorCapturePrey", E, {"predator": 4, "capture": 3, "n_agents": 7}) for E in (32768, 131072, 524288)]
            cases += [("Warehouse", E, {}) for E in (131072, 524288)]
            cases += [("Warehouse", E, {"n_agents": 6}) for E in (32768, 131072)]
            for scn, E, ov in cases:
                r = probe(scn, E, steps=100 if E < 100000 else 30, **ov)
                r["kernel"] = kern
                # mean over waves of the per-64-env maximum of the step's sweep count (TPE divergence cost)
                print(json.dumps(r), flush=True)
    if args.set == "rollout":   # K steps per launch (rg_rollout) vs K launches (rg_step)
        for kern in ("group", "tpe"):
            os.environ["RG_STEP_KERNEL"] = kern
            for E in ((4096, 32768) if kern == "group" else (32768, 524288)):
                env = VecRobotariumEnv("PredatorCapturePrey", E, overrides=OV["PredatorCapturePrey"])
                env.reset()
                for K in (1, 4, 16, 64):
                    if E * K > 524288 * 16:
                        continue
                    acts = torch.randint(0, 5, (K, E, 5), device=env.device, dtype=torch.int32)
                    buf = env.rollout(acts)
                    reps = max(4, 512 // K)
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(reps):
                        env.rollout(acts, out=buf)
                    b.record()
                    torch.cuda.synchronize()
                    us = a.elapsed_time(b) * 1e3 / (reps * K)
                    print(json.dumps({"kernel": kern, "E": E, "K": K, "us_per_step": us,
                                      "agent_steps_per_s": E * 5 / (us * 1e-6)}), flush=True)
    if args.set == "tpecap":   # thread-per-env kernel at a chip-filling batch: what the QP tail, replays and resets cost
        os.environ["RG_STEP_KERNEL"] = "tpe"
        for over, ar in (({}, True), ({"qp_max_sweeps": 4}, True), ({"qp_max_sweeps": 2}, True), ({"qp_max_sweeps": 1}, True),
                         ({"penalize_violations": False}, True), ({}, False)):
            r = probe("PredatorCapturePrey", 524288, steps=40, warm=40, auto_reset=ar, **over)
            print(json.dumps(r), flush=True)
    if args.set == "pairtest":   # headline / MaterialTransport / Warehouse launches and the multi-step launch, for A/B runs of
        # the shipped library against the -DRG_PROBE_NO_PAIRTEST build (ROBOGYM_LIB=marbler_amd/librobogym_nopair.so)
        os.environ["RG_STEP_KERNEL"] = "group"
        for scn, E in (("PredatorCapturePrey", 4096), ("MaterialTransport", 2048), ("MaterialTransport", 4096), ("Warehouse", 4096),
                       ("PredatorCapturePrey", 32768), ("MaterialTransport", 32768)):
            r = probe(scn, E, steps=400, warm=100)
            r["lib"] = os.environ.get("ROBOGYM_LIB", "shipped")
            print(json.dumps(r), flush=True)
        for scn, E, nact in (("PredatorCapturePrey", 4096, 5), ("MaterialTransport", 2048, 20)):
            env = VecRobotariumEnv(scn, E, overrides=OV[scn])
            env.reset()
            acts = torch.randint(0, nact, (64, E, env.N), device=env.device, dtype=torch.int32)
            buf = env.rollout(acts)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(8):
                env.rollout(acts, out=buf)
            b.record()
            torch.cuda.synchronize()
            print(json.dumps({"scenario": scn, "E": E, "api": "rg_rollout K=64", "us_per_step": a.elapsed_time(b) * 1e3 / (8 * 64),
                              "lib": os.environ.get("ROBOGYM_LIB", "shipped")}), flush=True)
    if args.set == "headline":   # the headline launch and its neighbours, for A/B runs of library builds (ROBOGYM_LIB)
        os.environ["RG_STEP_KERNEL"] = "group"
        for scn, E in (("PredatorCapturePrey", 4096), ("PredatorCapturePrey", 4096), ("PredatorCapturePrey", 2048), ("PredatorCapturePrey", 32768)):
            r = probe(scn, E, steps=1000, warm=200)
            r["lib"] = os.environ.get("ROBOGYM_LIB", "shipped")
            print(json.dumps(r), flush=True)
    if args.set == "big":     # one saturated configuration (for rocprofv3 --pmc runs); RG_STEP_KERNEL picks the kernel
        out.append(probe("PredatorCapturePrey", 524288, steps=20, warm=10))
    for r in out:
        print(json.dumps(r))
