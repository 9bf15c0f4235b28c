#!/usr/bin/env python3
"""One step of a golden fixture on the thread-per-env kernel of the -DRG_TPE_GUARD diagnostic build (stores through the
LDS staging block are bounds-checked: a store outside its array is dropped and flagged in done_count[0] instead of
faulting), compared with the float32 oracle.
    python -c "from marbler_amd import build; build.build(defines=('RG_TPE_GUARD',), out='marbler_amd/librobogym_guard.so')"
    python tests/guard_probe.py tests/golden/pcp_n6_capaware.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("ROBOGYM_LIB", os.path.join(ROOT, "marbler_amd", "librobogym_guard.so"))
os.environ["RG_STEP_KERNEL"] = "tpe"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # (this file lives in tests/: only tests may use the oracle)
import numpy as np
import torch
from helpers import gpu_from_state, load_golden, oracle_from_state, pre_state
from oracle import c_oracle
c_oracle.build_library()

for path in sys.argv[1:]:
    g, scenario, cfg = load_golden(path)
    state = pre_state(g)
    orc = oracle_from_state(c_oracle, scenario, cfg, state, np.float32)
    orc.step(g["actions"])
    env = gpu_from_state(scenario, cfg, state)
    env.done_count.zero_()
    a = torch.as_tensor(np.asarray(g["actions"], dtype=np.int32), device=env.device)
    obs, rew, done, info = env.step(a)
    torch.cuda.synchronize()
    flag = int(env.done_count[0].item())
    bad = int((obs.cpu().numpy().view(np.uint32) != orc.obs.view(np.uint32)).sum())
    post = {k: v.cpu().numpy() for k, v in env.state_dict().items()}
    others = {"reward": (rew.cpu().numpy(), orc.reward), "dist": (info["dist_travelled"].cpu().numpy(), orc.dist),
              "poses": (post["poses"], orc.poses), "carry": (post["carry_dist"], orc.carry)}
    for k, (x, y) in others.items():
        nb = int((x.view(np.uint32) != y.view(np.uint32)).sum())
        if nb:
            print(f"  {k}: {nb} words differ", flush=True)
            bad += nb
    print(f"{os.path.basename(path)}: guard flags 0x{flag:x}  words differing from the oracle (obs, reward, dist, poses, carry): {bad}", flush=True)
    if bad:
        go, oo = obs.cpu().numpy(), orc.obs
        idx = np.argwhere(go.view(np.uint32) != oo.view(np.uint32))
        print("  wrong words per env:", np.bincount(idx[:, 0], minlength=go.shape[0]).tolist())
        print("  wrong words per agent row:", np.bincount(idx[:, 1], minlength=go.shape[1]).tolist())
        print("  wrong words per column:", np.bincount(idx[:, 2], minlength=go.shape[2]).tolist())
        flat_o = oo.reshape(-1)
        for e, a_, c in idx[:12]:
            v = go[e, a_, c]
            hits = np.nonzero(flat_o.view(np.uint32) == v.view(np.uint32))[0][:4]
            print(f"  obs[{e},{a_},{c}] = {v!r}, oracle {oo[e, a_, c]!r}; the value is the oracle's flat index {hits.tolist()} (this one: {(e * go.shape[1] + a_) * go.shape[2] + c})")
    env.close()
