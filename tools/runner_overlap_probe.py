#!/usr/bin/env python3
"""Would two half batches on two streams collect faster than one batch?  The runner's time step is two latency-bound launches in a
dependent chain (actor -> env step); with the envs split in two halves, each on its own stream, one half's actor can run beside the
other half's env step.  Measures the in-place loop of BatchedRunner.run for 1 x E and 2 x E/2 (same kernels, same work).
    python tools/runner_overlap_probe.py [envs=4096] [T=300] [hidden=128]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from marbler_amd.evaluate import BatchedActor
from marbler_amd.gymma import GymmaVecEnv
from test_gpu_actor import _random_actor
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = "cuda:0"


class Half:
    def __init__(self, n, seed, stream):
        self.v = GymmaVecEnv("robotarium_gym:PredatorCapturePrey-v0", n, time_limit=1000, seed=seed)
        v = self.v
        self.actor = BatchedActor(_random_actor(1, v.obs_size + v.n_agents, H, v.n_actions, True, seed=2), v.n_agents, device=dev)
        self.stream = stream
        self.hidden = self.actor.init_hidden(n).to(dev)
        self.q = torch.empty(n, v.n_agents, v.n_actions, device=dev)
        self.obs = torch.zeros(T + 1, n, v.n_agents, v.obs_size, device=dev)
        self.actions = torch.empty(T, n, v.n_agents, dtype=torch.int32, device=dev)
        self.reward = torch.empty(T, n, device=dev)
        self.term = torch.zeros(T, n, dtype=torch.uint8, device=dev)
        self.restart = torch.ones(n, dtype=torch.uint8, device=dev)
        v.reset()
        torch.cuda.synchronize()
        v.env.set_stream(stream)

        # pointer-level form of the same two calls: no tensor views per step
        import ctypes as C
        from marbler_amd import _lib
        self.lib = _lib.load()
        self.ws = self.actor._weights_struct()
        self.actor.forward_fused(self.obs[0], self.hidden, restart=self.restart, q_out=self.q, actions_out=self.actions[0], stream=stream)
        torch.cuda.synchronize()
        self.sp = C.c_void_p(stream.cuda_stream)
        N, D = v.n_agents, v.obs_size
        self.args = [(self.obs[t].data_ptr(), (self.restart if t == 0 else self.term[t - 1]).data_ptr(), self.actions[t].data_ptr(),
                      self.obs[t + 1].data_ptr(), self.reward[t].data_ptr(), self.term[t].data_ptr()) for t in range(T)]
        self.n, self.N, self.D, self.hp, self.qp = n, N, D, self.hidden.data_ptr(), self.q.data_ptr()
        self.wsref = C.byref(self.ws)

    def step_ptr(self, t):
        o, r, a, o1, rw, tm = self.args[t]
        self.lib.rg_actor_forward(self.wsref, self.n, self.N, o, self.D, 1, r, self.hp, self.qp, a, self.sp)
        self.v.env.step_into(a, o1, rw, tm)

    def step(self, t):
        restart = self.restart if t == 0 else self.term[t - 1]
        self.actor.forward_fused(self.obs[t], self.hidden, restart=restart, q_out=self.q, actions_out=self.actions[t], stream=self.stream)
        rc = self.v.env.step_into(self.actions[t].data_ptr(), self.obs[t + 1].data_ptr(), self.reward[t].data_ptr(), self.term[t].data_ptr())
        assert rc == 0


def run(halves, ptr=False):
    steps = [h.step_ptr if ptr else h.step for h in halves]
    for t in range(20):
        for f in steps:
            f(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(T):
        for f in steps:
            f(t)
    host = (time.perf_counter() - t0) / T * 1e6
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / T * 1e6, host


s0 = torch.cuda.current_stream()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for ptr in (False, True):
    one = run([Half(E, 3, s0)], ptr)
    two = run([Half(E // 2, 3, sa), Half(E // 2, 4, sb)], ptr)
    print(f"E {E} hidden {H} {'pointer-level calls' if ptr else 'tensor-level calls'}: one batch {one[0]:.1f} us per time step (host loop {one[1]:.1f}); "
          f"two halves on two streams {two[0]:.1f} (host loop {two[1]:.1f})", flush=True)
