"""Out-of-bounds write detector: every device array of an env -- state, outputs, scratch, rg_rollout's [K, ...] outputs,
the gymma block -- is carved out of ONE slab with 1 KB red zones of a sentinel byte before and after it; after 50
auto-reset steps and a 16-step rg_rollout every red-zone byte must be intact.  The bit-exact comparisons elsewhere see a
stray store only when it lands in an array they read; torch's separate allocations hide one that lands in allocator slack.

Every scenario x agent count (2 .. 8, 12, 16 where the scenario admits it) x both step kernels x batch sizes that leave
a partly filled last wavefront in either mapping (1, 63, 64, 65, 4097), observation rows of every width class
(capability-aware on / off, more neighbours asked for than agents exist).  The outputs of the guarded run must also equal
those of an ordinary env (separate allocations, always the lane-group kernel) bit for bit.

Reference surface whose outputs these arrays are: wrapper.py:41-44.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SENTINEL = 0xA5
ZONE = 1024


def _guarded_class():
    import torch
    from marbler_amd import VecRobotariumEnv

    class GuardedEnv(VecRobotariumEnv):
        """All arrays from one slab, red zones between them."""

        def __init__(self, *a, slab_bytes=1 << 26, **kw):
            dev = torch.device(kw.get("device", "cuda:0"))
            self._slab = torch.full((slab_bytes,), SENTINEL, dtype=torch.uint8, device=dev)
            self._cursor = 0
            self._regions = []
            super().__init__(*a, **kw)

        def _alloc(self, shape, dtype, fill=0):
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            lo = (self._cursor + ZONE + 255) // 256 * 256
            hi = lo + nbytes
            if hi + ZONE > self._slab.numel():
                raise MemoryError("guard slab too small")
            self._cursor = hi
            self._regions.append((lo, hi))
            t = self._slab[lo:hi].view(dtype).view(tuple(shape))
            t.fill_(fill)
            return t

        def _alloc_outputs(self):   # one array each (the product packs them into one arena, 16-byte aligned, no gaps)
            E, N, D = self.E, self.N, self.D
            f32, i32, u8 = torch.float32, torch.int32, torch.uint8
            self.obs = self._alloc((E, N, D), f32)
            self.reward = self._alloc((E, N), f32)
            self.dist_travelled = self._alloc((E, N), f32)
            self.remaining = self._alloc((E,), i32, fill=-1)
            self.done_u8 = self._alloc((E,), u8)
            self.done = self.done_u8.view(torch.bool)
            self.violation = self._alloc((E,), u8)
            self._out_arena, self._out_offsets = None, {}

        def red_zones_intact(self):
            keep = torch.ones(self._slab.numel(), dtype=torch.bool, device=self._slab.device)
            for lo, hi in self._regions:
                keep[lo:hi] = False
            bad = torch.nonzero(keep & (self._slab != SENTINEL)).flatten()
            return bad.cpu().numpy()

        def owner_of(self, off):
            """Nearest array below a damaged byte (diagnostics)."""
            names = {}
            for k, v in self.__dict__.items():
                if isinstance(v, torch.Tensor) and v.device == self._slab.device and v.untyped_storage().data_ptr() == self._slab.untyped_storage().data_ptr():
                    names[v.data_ptr() - self._slab.data_ptr()] = k
            below = [o for o in names if o <= off]
            return names[max(below)] if below else "<slab start>"

    return GuardedEnv


def _configs():
    """(id, scenario, overrides, action count)"""
    out = []
    for n in (2, 3, 4, 5, 6, 7, 8, 12, 16):
        npred = (n + 1) // 2
        base = {"predator": npred, "capture": n - npred, "n_agents": n, "start_dist": 0.2 if n > 8 else 0.3}
        for cap in (False, True):
            out.append((f"pcp-n{n}-{'cap' if cap else 'plain'}", "PredatorCapturePrey", dict(base, capability_aware=cap), 5))
        # more neighbours asked for than agents exist: observation rows wider than the agent count fills
        out.append((f"pcp-n{n}-k{n}", "PredatorCapturePrey", dict(base, num_neighbors=n), 5))
        out.append((f"warehouse-n{n}", "Warehouse", {"n_agents": n, "start_dist": 0.4 if n > 8 else 0.6}, 5))
        out.append((f"warehouse-n{n}-k3", "Warehouse", {"n_agents": n, "start_dist": 0.4 if n > 8 else 0.6, "num_neighbors": 3}, 5))
        out.append((f"simple-n{n}", "Simple", {"n_agents": n, "start_dist": 0.2 if n > 8 else 0.3}, 5))
        if n >= 4:
            nf = n // 2
            mt = {"n_agents": n, "n_fast_agents": nf, "n_slow_agents": n - nf, "start_dist": 0.2 if n > 6 else 0.25}
            for cap in (False, True):
                out.append((f"mt-n{n}-{'cap' if cap else 'plain'}", "MaterialTransport", dict(mt, capability_aware=cap), 20))
    out.append(("arctic", "ArcticTransport", {}, 5))
    # the interior-point mode's kernels (their own instantiations, LDS workspaces, the gymma and rollout forms): N <= 8
    ip = {"barrier_solver": "cvxopt"}
    for n in (2, 5, 8):
        npred = (n + 1) // 2
        out.append((f"ipm-pcp-n{n}", "PredatorCapturePrey", dict(ip, predator=npred, capture=n - npred, n_agents=n), 5))
        out.append((f"ipm-warehouse-n{n}", "Warehouse", dict(ip, n_agents=n), 5))
    out.append(("ipm-mt-n6", "MaterialTransport", dict(ip, n_agents=6, n_fast_agents=3, n_slow_agents=3, start_dist=0.25), 20))
    out.append(("ipm-arctic", "ArcticTransport", dict(ip), 5))
    return out


CONFIGS = _configs()
SIZES = (1, 63, 64, 65, 4097)


@pytest.mark.parametrize("kernel", ["group", "tpe"])
@pytest.mark.parametrize("name,scenario,ov,n_act", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_no_store_outside_the_bound_arrays(name, scenario, ov, n_act, kernel, monkeypatch):
    import torch
    from marbler_amd import VecRobotariumEnv
    monkeypatch.setenv("RG_STEP_KERNEL", kernel)
    Guarded = _guarded_class()
    STEPS, K = 50, 16
    for E in SIZES:
        with monkeypatch.context() as m:   # the unguarded twin always runs the lane-group kernel: the two mappings must agree
            m.setenv("RG_STEP_KERNEL", "group")
            ref = VecRobotariumEnv(scenario, E, overrides=ov, seed=7, collect_qp_stats=True)
        slab = int(E * ref.N * ref.D * 4 * (K + 1) * 1.25) + E * 8192 + (4 << 20)
        env = Guarded(scenario, E, overrides=ov, seed=7, collect_qp_stats=True, slab_bytes=slab)
        g = torch.Generator(device=env.device)
        g.manual_seed(E)
        env.reset()
        ref.reset()
        assert env.red_zones_intact().size == 0, f"{name} E={E}: rg_reset wrote outside its arrays"
        acts = torch.randint(0, n_act, (STEPS + K, E, env.N), generator=g, device=env.device, dtype=torch.int32)
        for t in range(STEPS):
            o1, r1, d1, i1 = env.step(acts[t])
            o2, r2, d2, i2 = ref.step(acts[t])
            assert torch.equal(o1.view(torch.int32), o2.view(torch.int32)), (name, E, t)
            assert torch.equal(r1.view(torch.int32), r2.view(torch.int32)) and torch.equal(d1, d2), (name, E, t)
            for key in i1:
                assert torch.equal(i1[key], i2[key]), (name, E, t, key)
        bad = env.red_zones_intact()
        assert bad.size == 0, (f"{name} E={E} kernel={kernel}: rg_step damaged {bad.size} red-zone bytes, first at slab offset "
                               f"{int(bad[0])} (after `{env.owner_of(int(bad[0]))}`)")
        if True:   # (every shape since round 4: rg_rollout no longer asks for E*N*D % 4 == 0)
            out1 = env.rollout(acts[STEPS:])
            out2 = ref.rollout(acts[STEPS:])
            for key in ("obs", "reward", "done", "dist_travelled", "violation", "remaining"):
                assert torch.equal(out1[key], out2[key]), (name, E, "rollout", key)
            bad = env.red_zones_intact()
            assert bad.size == 0, (f"{name} E={E} kernel={kernel}: rg_rollout damaged {bad.size} red-zone bytes, first at slab "
                                   f"offset {int(bad[0])} (after `{env.owner_of(int(bad[0]))}`)")
        o1 = env.get_obs(out=env.obs)
        assert torch.equal(o1, ref.get_obs())
        assert env.red_zones_intact().size == 0, f"{name} E={E}: rg_get_obs wrote outside its array"
        sa, sb = env.state_dict(), ref.state_dict()
        for key in sa:
            assert torch.equal(sa[key], sb[key]), (name, E, key)
        env.close()
        ref.close()


@pytest.mark.parametrize("kernel", ["group", "tpe"])
def test_gymma_block_stays_inside_its_arrays(kernel, monkeypatch):
    """The fused TimeLimit / reduction outputs (rg_step_io's gymma block)."""
    import torch
    monkeypatch.setenv("RG_STEP_KERNEL", kernel)
    Guarded = _guarded_class()
    for scenario, ov, n_act in (("PredatorCapturePrey", {"predator": 3, "capture": 2, "n_agents": 5}, 5),
                                ("Warehouse", {"n_agents": 8}, 5), ("MaterialTransport", {}, 20), ("ArcticTransport", {}, 5)):
        for E in (1, 65, 4097):
            env = Guarded(scenario, E, overrides=ov, seed=3)
            env.enable_time_limit(9)
            env.reset()
            g = torch.Generator(device=env.device)
            g.manual_seed(1)
            for t in range(30):
                env.step(torch.randint(0, n_act, (E, env.N), generator=g, device=env.device, dtype=torch.int32))
            assert int(env.truncated.sum()) >= 0
            bad = env.red_zones_intact()
            assert bad.size == 0, f"{scenario} E={E}: damaged red zone after `{env.owner_of(int(bad[0]))}`"
            env.close()


def test_the_detector_detects():
    """A deliberate one-byte store just past an array must be reported (the test of the test)."""
    import torch
    Guarded = _guarded_class()
    env = Guarded("Warehouse", 65, overrides={}, seed=0)
    assert env.red_zones_intact().size == 0
    off = env.reward.data_ptr() - env._slab.data_ptr() + env.reward.numel() * 4
    env._slab[off] = 0
    bad = env.red_zones_intact()
    assert bad.size == 1 and int(bad[0]) == off and env.owner_of(off) == "reward"
    env._slab[off] = SENTINEL
    lo = env.poses.data_ptr() - env._slab.data_ptr() - 1
    env._slab[lo] = 1
    assert env.red_zones_intact().tolist() == [lo]
    env.close()
