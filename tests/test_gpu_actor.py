"""The fused actor kernel (csrc/actor_mfma.hip, f32 MFMA) against the torch float32 evaluation of the
same weights (marbler_amd.evaluate.BatchedActor.forward, itself pinned to the reference's RNNAgent /
RNNNSAgent modules by tests/test_evaluate.py): action values and hidden state within 1e-5, greedy
actions equal wherever the top two action values are more than 1e-4 apart."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _random_actor(n_sets, I, H, A, use_rnn, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * 0.3  # noqa: E731
    sd = {}
    for i in range(n_sets):
        pre = f"agents.{i}." if n_sets > 1 else ""
        sd[pre + "fc1.weight"], sd[pre + "fc1.bias"] = r(H, I), r(H)
        if use_rnn:
            sd[pre + "rnn.weight_ih"], sd[pre + "rnn.weight_hh"] = r(3 * H, H), r(3 * H, H)
            sd[pre + "rnn.bias_ih"], sd[pre + "rnn.bias_hh"] = r(3 * H), r(3 * H)
        else:
            sd[pre + "rnn.weight"], sd[pre + "rnn.bias"] = r(H, H), r(H)
        sd[pre + "fc2.weight"], sd[pre + "fc2.bias"] = r(A, H), r(A)
    return sd


@pytest.mark.parametrize("shared,H,E,N,D,A,use_rnn,append", [
    (True, 128, 300, 5, 16, 5, True, True),      # the PredatorCapturePrey model zoo shape (qmix.json)
    (True, 128, 70, 5, 16, 5, True, None),       # the same on torch-layout GRU weights (no packing: f32-input MFMA)
    (True, 128, 70, 5, 16, 5, True, "f32"),      # ... and on the float32 streaming order of rg_actor_pack_gru (f32-input MFMA)
    (False, 128, 40, 4, 30, 5, True, "f32"),
    (True, 64, 77, 4, 9, 20, True, True),        # MaterialTransport: 20 actions
    (False, 64, 130, 5, 16, 5, True, False),     # rnn_ns: one network per agent, no agent id
    (True, 128, 33, 8, 18, 5, False, True),      # use_rnn = False
    (False, 128, 1, 3, 30, 5, True, False),      # a single env, ragged tile
    (False, 128, 50, 4, 9, 20, False, False),    # the zoo's MaterialTransport mappo_ns: one MLP-layer network per agent (round 4: its weight stride)
    (True, 128, 65, 4, 28, 5, True, True),       # input width exactly 32: the widest layer fc1 stages through LDS
    (False, 64, 65, 3, 32, 7, True, False),      # ... per-agent weights, no agent id, two wavefronts per tile
    (True, 64, 65, 4, 29, 5, True, True),        # input width 33: the first one fc1 reads straight from memory
    (False, 128, 35, 2, 62, 3, True, True),      # input width 64: the widest supported
    (True, 128, 35, 3, 1, 2, True, False),       # input width 1
])
def test_fused_actor_matches_torch(shared, H, E, N, D, A, use_rnn, append):
    from marbler_amd.evaluate import BatchedActor
    dev = "cuda:0"
    pack = {None: False, "f32": "f32"}.get(append, True)     # True: the default, three bfloat16 planes (rg_actor_pack_gru_bf16x3)
    append = True if append in (None, "f32") else append
    I = D + (N if append else 0)
    sd = _random_actor(1 if shared else N, I, H, A, use_rnn, seed=H + E)
    actor = BatchedActor(sd, N, use_rnn=use_rnn, device=dev, pack_gru=pack)
    assert actor.fused_supported()
    g = torch.Generator(device=dev).manual_seed(1)
    hidden = (torch.rand(E, N, H, generator=g, device=dev) * 2 - 1)
    eye = torch.eye(N, device=dev).unsqueeze(0).expand(E, N, N)
    h_ref = hidden.clone()
    for step in range(3):   # the hidden state feeds back
        obs = torch.rand(E, N, D, generator=g, device=dev) * 3 - 1.5
        restart = (torch.rand(E, generator=g, device=dev) < 0.2).to(torch.uint8)
        fresh = restart.bool()[:, None, None]      # a new episode: zero observation, zero hidden state
        obs_seen = torch.where(fresh, torch.zeros_like(obs), obs)
        inp = torch.cat([obs_seen, eye], dim=2) if append else obs_seen
        h_in = torch.where(restart.bool()[:, None, None], torch.zeros_like(h_ref), h_ref)
        q_ref, h_ref = actor.forward(inp, h_in)
        q, act = actor.forward_fused(obs, hidden, append_agent_id=append, restart=restart)
        torch.cuda.synchronize()
        assert float((q - q_ref).abs().max()) < 1e-5, step
        assert float((hidden - h_ref).abs().max()) < 1e-5, step
        top2 = q_ref.topk(2, dim=2).values
        clear = (top2[..., 0] - top2[..., 1]) > 1e-4
        assert torch.equal(act[clear].long(), q_ref.argmax(dim=2)[clear])
        assert bool(clear.float().mean() > 0.9)


@pytest.mark.parametrize("name", ["actor_shared_gru_h64", "actor_ns_gru_h64"])
def test_fused_actor_on_the_reference_vectors(name):
    """Golden vectors from the reference's own RNNAgent / RNNNSAgent (tests/golden/make_actor_golden.py)."""
    from marbler_amd.evaluate import BatchedActor
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    sd = {k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("sd_")}
    T, N, I = g["inputs"].shape
    actor = BatchedActor(sd, N, use_rnn=bool(g["use_rnn"]), device="cuda:0")
    E = 3                                            # the same env three times: batching must not mix rows
    hidden = actor.init_hidden(E)
    for t in range(T):
        x = torch.as_tensor(g["inputs"][t]).unsqueeze(0).expand(E, N, I).contiguous().cuda()   # carries the agent id already
        q, _ = actor.forward_fused(x, hidden, append_agent_id=False)
        for e in range(E):
            assert np.abs(q[e].cpu().numpy() - g["q"][t]).max() < 1e-5
            assert np.abs(hidden[e].cpu().numpy() - g["h"][t]).max() < 1e-5


def _draw_actor_case(rng):
    H = int(rng.choice([64, 128]))
    shared = bool(rng.rand() < 0.6)
    N = int(rng.randint(1, 9))
    append = bool(rng.rand() < 0.6)
    D = int(rng.randint(1, 64 - (N if append else 0) + 1))          # input width D (+ N) <= 64
    A = int(rng.choice([1, 2, 5, 7, 20, 31, 32]))
    E = int(rng.choice([1, 2, 31, 32, 33, 100, 257]))
    use_rnn = bool(rng.rand() < 0.8)
    pack = [True, "f32", False][int(rng.randint(0, 3))]
    if rng.rand() < 0.4:     # (drawn after everything else: the cases of earlier rounds keep their shapes)
        pack = "f16x2" if pack is True else "bf16x3" if pack == "f32" else pack
    return shared, H, E, N, D, A, use_rnn, append, pack


@pytest.mark.parametrize("seed", range(96))
def test_fused_actor_random_shapes(seed):
    """Shapes nobody listed: every input width up to 64 (ragged fc1 rows, agent id on or off), 1 ... 32 actions (fc2's padded tile,
    the all-thread arg-max), ragged and tiny batches, shared / per-agent weights, GRU in all three weight forms / the MLP layer --
    against the torch evaluation: 1e-5 on q and hidden, greedy actions equal where the top two values are 1e-4 apart."""
    from marbler_amd.evaluate import BatchedActor
    rng = np.random.RandomState(7000 + seed)
    shared, H, E, N, D, A, use_rnn, append, pack = _draw_actor_case(rng)
    dev = "cuda:0"
    I = D + (N if append else 0)
    actor = BatchedActor(_random_actor(1 if shared else N, I, H, A, use_rnn, seed), N, use_rnn=use_rnn, device=dev, pack_gru=pack)
    g = torch.Generator(device=dev).manual_seed(seed)
    hidden = torch.rand(E, N, H, generator=g, device=dev) * 2 - 1
    h_ref = hidden.clone()
    eye = torch.eye(N, device=dev).unsqueeze(0).expand(E, N, N)
    for step in range(2):
        obs = torch.rand(E, N, D, generator=g, device=dev) * 3 - 1.5
        restart = (torch.rand(E, generator=g, device=dev) < 0.3).to(torch.uint8)
        fresh = restart.bool()[:, None, None]
        obs_seen = torch.where(fresh, torch.zeros_like(obs), obs)
        q_ref, h_ref = actor.forward(torch.cat([obs_seen, eye], dim=2) if append else obs_seen, torch.where(fresh, torch.zeros_like(h_ref), h_ref))
        q, act = actor.forward_fused(obs, hidden, append_agent_id=append, restart=restart)
        torch.cuda.synchronize()
        case = (shared, H, E, N, D, A, use_rnn, append, pack, step)
        assert float((q - q_ref).abs().max()) < 1e-5, case
        assert float((hidden - h_ref).abs().max()) < 1e-5, case
        if A > 1:
            top2 = q_ref.topk(2, dim=2).values
            clear = (top2[..., 0] - top2[..., 1]) > 1e-4
            assert torch.equal(act[clear].long(), q_ref.argmax(dim=2)[clear]), case
        else:
            assert int(act.abs().sum()) == 0, case


@pytest.mark.parametrize("A", [1, 5, 20, 32])
def test_greedy_action_is_a_valid_index_on_non_finite_rows(A):
    """Rows whose action values are all NaN, all -inf, or mixed: the greedy action is what torch.argmax gives (a NaN counts as the
    largest value, the first one wins; an all -inf row gives column 0) and ALWAYS an index in [0, A) -- the batched runner feeds it
    straight back into env.step and into gathers.  The non-finite values are put in through fc2's bias (per agent: non-shared
    weights), so every other stage of the kernel runs on ordinary numbers."""
    from marbler_amd.evaluate import BatchedActor
    dev = "cuda:0"
    N, H, D = 6, 64, 7
    sd = _random_actor(N, D, H, A, True, 5)
    b2 = [k for k in sd if k.endswith("fc2.bias")]
    assert len(b2) == N, sorted(sd)
    nan, ninf = float("nan"), float("-inf")
    for a, k in enumerate(sorted(b2)):
        b = sd[k].clone()
        if a == 0:
            b[:] = nan                       # every column NaN
        elif a == 1:
            b[:] = ninf                      # every column -inf
        elif a == 2 and A > 2:
            b[A // 2] = nan                  # one NaN among finite values: it wins
        elif a == 3 and A > 2:
            b[:] = ninf
            b[A - 1] = -3.2e38               # below the old sentinel -3.0e38, but the only finite value
        elif a == 4 and A > 3:
            b[1] = nan
            b[A - 2] = nan                   # two NaNs: the first
        sd[k] = b
    actor = BatchedActor(sd, N, use_rnn=True, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    E = 70
    hidden = torch.rand(E, N, H, generator=g, device=dev) * 2 - 1
    obs = torch.rand(E, N, D, generator=g, device=dev)
    q_ref, _ = actor.forward(obs, hidden.clone())
    q, act = actor.forward_fused(obs, hidden, append_agent_id=False)
    torch.cuda.synchronize()
    assert int(act.min()) >= 0 and int(act.max()) < A
    assert torch.equal(torch.isnan(q), torch.isnan(q_ref))
    want = q_ref.argmax(dim=2)
    for a in range(5):                       # the rows made non-finite: exactly torch's answer
        assert torch.equal(act[:, a].long(), want[:, a]), (a, act[:3, a], want[:3, a])


@pytest.mark.parametrize("shared,H,A,eps", [(True, 128, 5, 0.25), (False, 64, 20, 0.05), (True, 64, 5, 1.0), (True, 64, 32, 1e-6)])
def test_epsilon_greedy_selection_inside_the_launch(shared, H, A, eps):
    """rg_actor_forward_explore: the same launch with EPyMARL's epsilon-greedy selection folded in.  Everything but the actions is
    identical to the plain launch; the actions equal the rule in torch ops (evaluate.explore_select) applied to the plain launch's
    greedy actions and the same uniforms; the explored share is epsilon and the explored actions are uniform."""
    from marbler_amd import _lib
    from marbler_amd.evaluate import BatchedActor, explore_select
    dev = "cuda:0"
    E, N, D = 3000, 4, 9
    actor = BatchedActor(_random_actor(1 if shared else N, D + N, H, A, True, 8), N, use_rnn=True, device=dev)
    g = torch.Generator(device=dev).manual_seed(11)
    hidden = torch.rand(E, N, H, generator=g, device=dev) * 2 - 1
    obs = torch.rand(E, N, D, generator=g, device=dev)
    u = torch.rand(E, N, generator=g, device=dev)
    u[0, 0], u[0, 1] = 0.0, float(np.nextafter(np.float32(1.0), np.float32(0.0)))    # the ends of the range
    h0, h1 = hidden.clone(), hidden.clone()
    q0, greedy = actor.forward_fused(obs, h0)
    q1, act = actor.forward_fused(obs, h1, explore_u=u, epsilon=eps)
    torch.cuda.synchronize()
    assert torch.equal(q0, q1) and torch.equal(h0, h1)
    want = explore_select(greedy, u, eps, A)
    assert torch.equal(act, want)
    assert int(act.min()) >= 0 and int(act.max()) < A
    assert int(act[0, 0]) == 0 and (eps == 1.0 or int(act[0, 1]) == int(greedy[0, 1]))
    explored = u < eps
    if eps >= 0.05:
        share = float(explored.float().mean())
        assert abs(share - eps) < 4 * (eps * (1 - eps) / (E * N)) ** 0.5 + 1e-3
        counts = torch.bincount(act[explored].long(), minlength=A).float()
        expect = float(explored.sum()) / A
        assert float(((counts - expect) ** 2 / expect).sum()) < 3 * A + 20      # chi-square, loose
    # what is not explored is the greedy action (the boundary u ~ epsilon is decided by the float32 product: skip one ulp of it)
    clear = (u >= eps * 1.0001) | (u <= eps * 0.9999)
    assert torch.equal(act[clear & ~explored], greedy[clear & ~explored])
    assert bool((act[clear & explored] == (u[clear & explored] / eps * A).floor().clamp(max=A - 1).int()).float().mean() > 0.999)
    # argument checks
    with pytest.raises(_lib.RobogymError):
        actor.forward_fused(obs, h1, explore_u=u, epsilon=0.0)
    with pytest.raises(ValueError):
        actor.forward_fused(obs, h1, explore_u=u[:, :2], epsilon=eps)


@pytest.mark.parametrize("shared,H,E,N,D,A", [(True, 128, 300, 5, 16, 5), (False, 64, 130, 5, 16, 5), (True, 64, 77, 4, 9, 20),
                                                (False, 128, 33, 3, 30, 7)])
def test_two_binary16_planes_match_torch_and_the_three_plane_form(shared, H, E, N, D, A):
    """gru_packed = 3 (rg_actor_pack_gru_f16x2): three plane products per float32 product.  Against torch's float32 evaluation to the
    same 1e-5 as every other form over a few recurrent steps, and against the three-bfloat16-plane form (error far below float32's
    own) to 3e-6 -- a float32 GEMM's own roundings are of that size."""
    from marbler_amd.evaluate import BatchedActor
    dev = "cuda:0"
    sd = _random_actor(1 if shared else N, D + N, H, A, True, seed=H + E)
    fast, fine = (BatchedActor(sd, N, use_rnn=True, device=dev, pack_gru=p) for p in ("f16x2", "bf16x3"))
    g = torch.Generator(device=dev).manual_seed(2)
    hidden = torch.rand(E, N, H, generator=g, device=dev) * 2 - 1
    h_fine, h_ref = hidden.clone(), hidden.clone()
    eye = torch.eye(N, device=dev).unsqueeze(0).expand(E, N, N)
    worst = 0.0
    for step in range(4):
        obs = torch.rand(E, N, D, generator=g, device=dev) * 3 - 1.5
        q_ref, h_ref = fast.forward(torch.cat([obs, eye], dim=2), h_ref)
        q, act = fast.forward_fused(obs, hidden)
        q3, act3 = fine.forward_fused(obs, h_fine)
        torch.cuda.synchronize()
        assert float((q - q_ref).abs().max()) < 1e-5 and float((hidden - h_ref).abs().max()) < 1e-5, step
        worst = max(worst, float((q - q3).abs().max()), float((hidden - h_fine).abs().max()))
        top2 = q_ref.topk(2, dim=2).values
        clear = (top2[..., 0] - top2[..., 1]) > 1e-4
        assert torch.equal(act[clear], act3[clear])
    assert worst < 3e-6, worst


def test_two_binary16_planes_on_values_outside_binary16s_normal_range():
    """Activations and weights below 2^-14 (binary16's smallest normal number) travel in the scaled low plane alone; the result must
    not lose them: a network whose recurrent weights and hidden state are mostly tiny, against the three-plane form."""
    from marbler_amd.evaluate import BatchedActor
    dev = "cuda:0"
    E, N, D, H, A = 64, 4, 12, 128, 5
    sd = _random_actor(1, D + N, H, A, True, seed=21)
    g0 = torch.Generator().manual_seed(4)
    for k in ("rnn.weight_hh", "rnn.weight_ih"):
        w = sd[k]
        tiny = torch.rand(w.shape, generator=g0) < 0.5
        scale = 10.0 ** (-4 - 4 * torch.rand(w.shape, generator=g0))       # 1e-4 ... 1e-8
        sd[k] = torch.where(tiny, w * scale, w)
    fast, fine = (BatchedActor(sd, N, use_rnn=True, device=dev, pack_gru=p) for p in ("f16x2", "bf16x3"))
    g = torch.Generator(device=dev).manual_seed(6)
    for amp in (1.0, 3e-5, 1e-7):
        hidden = (torch.rand(E, N, H, generator=g, device=dev) * 2 - 1) * amp
        # large recurrent products on purpose: tiny h x weights scaled UP would show a lost hi plane
        obs = torch.rand(E, N, D, generator=g, device=dev) * amp
        h2 = hidden.clone()
        q, _ = fast.forward_fused(obs, hidden)
        q3, _ = fine.forward_fused(obs, h2)
        torch.cuda.synchronize()
        assert float((q - q3).abs().max()) < 2e-6 and float((hidden - h2).abs().max()) < 2e-6, amp
    # the lost-plane check proper: gh = h W_hh with h ~ 3e-5 and W_hh ~ 3e3 (products of order 1 from operands below 2^-14)
    sd2 = dict(sd)
    sd2["rnn.weight_hh"] = sd["rnn.weight_hh"].sign() * 3e3 * torch.rand(sd["rnn.weight_hh"].shape, generator=g0)
    fast, fine = (BatchedActor(sd2, N, use_rnn=True, device=dev, pack_gru=p) for p in ("f16x2", "bf16x3"))
    hidden = (torch.rand(E, N, H, generator=g, device=dev) * 2 - 1) * 3e-5
    obs = torch.rand(E, N, D, generator=g, device=dev)
    h2 = hidden.clone()
    q, _ = fast.forward_fused(obs, hidden)
    q3, _ = fine.forward_fused(obs, h2)
    torch.cuda.synchronize()
    assert float((q - q3).abs().max()) < 1e-5 and float((hidden - h2).abs().max()) < 1e-5


def test_two_binary16_planes_saturate_beyond_binary16s_range_and_keep_nans():
    """An activation beyond 65 504 (here: a hidden unit of fc1 whose bias is 1e5, its weights zero) would convert to infinity and its
    low plane to NaN; the two-plane form saturates it instead -- the outputs are finite and bit-identical to those of the same
    network with that bias AT 65 504 -- and a NaN observation still gives NaN action values, in its own rows only."""
    from marbler_amd.evaluate import BatchedActor
    dev = "cuda:0"
    E, N, D, H, A = 40, 4, 12, 128, 5
    outs = []
    for bias in (1e5, 65504.0):
        sd = _random_actor(1, D + N, H, A, True, seed=31)
        sd["fc1.weight"][3] = 0.0
        sd["fc1.bias"][3] = bias
        sd["rnn.weight_ih"][:, 3] *= 1e-3          # (keeps the gates' pre-activations in a range where the outputs still move)
        actor = BatchedActor(sd, N, use_rnn=True, device=dev, pack_gru="f16x2")
        g = torch.Generator(device=dev).manual_seed(8)
        hidden = torch.rand(E, N, H, generator=g, device=dev) * 2 - 1
        obs = torch.rand(E, N, D, generator=g, device=dev)
        q, act = actor.forward_fused(obs, hidden)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(q).all()) and bool(torch.isfinite(hidden).all()), bias
        outs.append((q.clone(), hidden.clone(), act.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    obs[5, 2, 0] = float("nan")
    hidden = torch.rand(E, N, H, generator=g, device=dev) * 2 - 1
    q, act = actor.forward_fused(obs, hidden)
    torch.cuda.synchronize()
    bad = torch.isnan(q).any(dim=2)
    assert bool(bad[5, 2]) and int(bad.sum()) == 1
    assert int(act.min()) >= 0 and int(act.max()) < A
