"""masked categorical action selection (acktr/distributions.py:71-84, acktr/model.py:56-68).
This is the one floating-point kernel of the repository, so its checker is a plain PyTorch float32
reference of the same op.  Tolerances: log-probabilities |diff| <= 5e-6 (float32 sums of up to 512 terms
in a different order: sequential in the oracle, tree-shaped on the GPU, vectorised in torch); the deterministic action must equal torch's argmax whenever the top-2 probabilities
differ by more than 1e-6; a sampled action must sit where the float64 CDF crosses u*total within 1e-5."""
import numpy as np
import pytest
import torch


def torch_reference(x, m):
    lx = torch.softmax(x - (1.0 - m) * 14.0, dim=-1) + 1e-5          # distributions.py:76-80
    return torch.distributions.Categorical(probs=lx)                  # FixedCategorical(probs=lx)


def make_case(E, M, seed, density=0.3):
    rng = np.random.RandomState(seed)
    x = (rng.randn(E, M) * rng.choice([0.1, 1.0, 4.0], size=(E, 1))).astype(np.float32)
    m = (rng.rand(E, M) < density).astype(np.float32)
    m[0] = 0.0                                                         # nothing feasible
    m[1] = 1.0                                                         # everything feasible
    return x, m


def uniform_of(seed, gid, step):
    """The counter-based stream of include/bpp_abi.h: 32-bit multiply-xorshift hash, top 24 bits."""
    m32 = 0xFFFFFFFF
    h = ((seed & m32) ^ (((seed >> 32) * 0x9E3779B1) & m32)) ^ ((((step & m32) + (step >> 32) * 0xC2B2AE3D) * 0x27D4EB2F) & m32)
    h ^= (gid * 0x85EBCA77) & m32
    h ^= h >> 16
    h = (h * 0x7FEB352D) & m32
    h ^= h >> 15
    h = (h * 0x846CA68B) & m32
    h ^= h >> 16
    return (h >> 8) / 16777216.0


def check(act_fn, E, M, seed):
    x, m = make_case(E, M, seed)
    d = torch_reference(torch.from_numpy(x), torch.from_numpy(m))
    probs = d.probs.numpy()
    # deterministic
    a, lp = act_fn(x, m, 5, 9, True)
    top2 = np.sort(probs, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-6
    np.testing.assert_array_equal(a[clear], probs.argmax(1)[clear])
    np.testing.assert_allclose(lp, d.log_prob(torch.from_numpy(a)).numpy(), rtol=0, atol=5e-6)
    # sampled
    a, lp = act_fn(x, m, 5, 9, False)
    np.testing.assert_allclose(lp, d.log_prob(torch.from_numpy(a)).numpy(), rtol=0, atol=5e-6)
    cdf = np.cumsum(probs.astype(np.float64), axis=1)
    for e in range(E):
        u = uniform_of(5, e, 9)
        lo = cdf[e, a[e] - 1] if a[e] > 0 else 0.0
        assert lo - 1e-5 <= u <= cdf[e, a[e]] + 1e-5, (e, a[e], u, lo, cdf[e, a[e]])
    return x, m, probs


@pytest.mark.parametrize("E,M", [(300, 100), (130, 200), (50, 400), (64, 8)])
def test_oracle_masked_act_matches_torch_reference(oracle, E, M):
    check(lambda x, m, s, t, det: oracle.masked_act(x, m, s, t, det), E, M, seed=E + M)


def test_oracle_masked_act_sampling_frequencies(oracle):
    """Many draws of one row follow its distribution (chi-square-ish bound on every bucket)."""
    rng = np.random.RandomState(3)
    M, N = 100, 40000
    x = np.tile((rng.randn(1, M) * 1.5).astype(np.float32), (N, 1))
    m = np.tile((rng.rand(1, M) < 0.4).astype(np.float32), (N, 1))
    a, _ = oracle.masked_act(x, m, 11, 0, False)
    p = torch_reference(torch.from_numpy(x[:1]), torch.from_numpy(m[:1])).probs.numpy()[0]
    freq = np.bincount(a, minlength=M) / N
    assert np.abs(freq - p).max() < 4 * np.sqrt(p.max() / N) + 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("E,M", [(4099, 100), (1000, 200), (333, 400), (40, 8), (70, 512), (129, 800), (33, 7), (50, 1023)])
def test_gpu_masked_act_matches_torch_reference_and_oracle(oracle, E, M):
    import bpp_amd

    def gpu(x, m, s, t, det):
        a, lp = bpp_amd.masked_act(torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda(), s, t, det)
        assert a.dtype == torch.int64 and tuple(a.shape) == (x.shape[0], 1) and tuple(lp.shape) == (x.shape[0], 1)
        return a.cpu().numpy()[:, 0], lp.cpu().numpy()[:, 0]

    x, m, probs = check(gpu, E, M, seed=E + M)
    # HIP vs the C restatement: same draws except where float32 summation order moves a CDF crossing
    for det in (True, False):
        ag, lg = gpu(x, m, 5, 9, det)
        ao, lo = oracle.masked_act(x, m, 5, 9, det)
        assert (ag != ao).mean() < 0.01
        # the oracle sums sequentially in float32: its own error grows with M (the torch reference above is the
        # authority for this kernel); 5e-6 up to 512 entries, 2.5e-8 per entry beyond
        np.testing.assert_allclose(lg[ag == ao], lo[ag == ao], rtol=0, atol=max(5e-6, 2.5e-8 * M))


def test_oracle_masked_act_against_the_reference_policy_head(oracle):
    """Build container only: the REFERENCE's own Policy (acktr/model.py) built on this package's spaces,
    fed observations/masks produced by the oracle env; its `Categorical.forward` + `mode()` + `log_probs()`
    (acktr/distributions.py:71-84, acktr/model.py:56-68) are what bpp_masked_act must reproduce."""
    from oracle import ref_shims
    if not ref_shims.available():
        pytest.skip("reference tree not present")
    import types
    import bpp_amd
    ref_shims.install()
    from acktr.model import Policy
    from acktr.storage import RolloutStorage
    for rot in (False, True):
        size, E = (10, 10, 10), 64
        args = types.SimpleNamespace(channel=4, container_size=size, pallet_size=10, enable_rotation=rot)
        obs_space, act_space = bpp_amd.Box(0.0, 10, (400,)), bpp_amd.Discrete(100 * (1 + rot))
        torch.manual_seed(0)
        policy = Policy(obs_space.shape, act_space, base_kwargs={"recurrent": False, "hidden_size": 256, "args": args})
        RolloutStorage(5, E, obs_space.shape, act_space, policy.recurrent_hidden_state_size, can_give_up=False,
                       enable_rotation=rot, pallet_size=10)                       # accepts the spaces as well
        env = oracle.OracleEnv(bpp_amd.sequences.cut2_pool(size, 16, seed=0), size, rot, E)
        obs, mask = env.reset()
        for t in range(6):
            a = oracle.sample_feasible(mask, 3, t)
            o = env.step(a)
            obs, mask = o["obs"], o["mask"]
        with torch.no_grad():
            ot, mt = torch.from_numpy(obs), torch.from_numpy(mask)
            value, action, logp, _ = policy.act(ot, None, None, mt, deterministic=True)
            _, features, _, _ = policy.base(ot, None, None)
            logits = policy.dist.linear(features).numpy() * 50.0       # the head's own logits, spread out
            dist, _, _ = policy.dist(features, mt)                     # un-scaled: the reference's own path
            a0, lp0 = oracle.masked_act(policy.dist.linear(features).numpy(), mask, 0, 0, True)
            np.testing.assert_allclose(lp0, dist.log_probs(torch.from_numpy(a0).unsqueeze(1)).numpy()[:, 0], atol=5e-6)
            top2 = np.sort(dist.probs.numpy(), 1)[:, -2:]
            clear = (top2[:, 1] - top2[:, 0]) > 1e-6
            np.testing.assert_array_equal(a0[clear], action.numpy()[clear, 0])
            # spread-out logits: compare against the reference formula applied to them
            lx = torch.softmax(torch.from_numpy(logits) - (1 - mt) * 14, -1) + 1e-5
            d2 = torch.distributions.Categorical(probs=lx)
            a1, lp1 = oracle.masked_act(logits, mask, 5, 1, False)
            np.testing.assert_allclose(lp1, d2.log_prob(torch.from_numpy(a1)).numpy(), atol=5e-6)
            assert (mask[np.arange(E), a1] == 1).mean() > 0.95          # draws land on feasible positions


def test_oracle_masked_act_counter_equals_by_value(oracle):
    """bpp_masked_act_counter (ABI v16): (seed, step) handed over in memory give the draws of the by-value entry point."""
    x, m = make_case(200, 100, 3)
    for det in (False, True):
        a0, l0 = oracle.masked_act(x, m, 11, 42, det)
        a1, l1 = oracle.masked_act(x, m, 11, 42, det, counter=True)
        np.testing.assert_array_equal(a0, a1)
        np.testing.assert_array_equal(l0, l1)
    assert (oracle.masked_act(x, m, 11, 43, False, counter=True)[0] != a0).any()


def test_emulated_masked_act_counter_equals_by_value(emu):
    x, m = make_case(70, 100, 4)
    for M in (100, 7):
        for det in (False, True):
            a0, l0 = emu.masked_act(x[:, :M], m[:, :M], 5, 9, det)
            a1, l1 = emu.masked_act(x[:, :M], m[:, :M], 5, 9, det, counter=True)
            np.testing.assert_array_equal(a0, a1)
            np.testing.assert_array_equal(l0, l1)


@pytest.mark.gpu
@pytest.mark.parametrize("E,rot", [(64, False), (1024, True)])
def test_gpu_lock_step_captured_in_a_hip_graph_equals_the_eager_loop(E, rot):
    """A whole lock-step -- a policy forward, bpp_masked_act_counter (its (seed, step) in device memory), the fused environment
    step, the counter's increment -- captured ONCE in a HIP graph and replayed 37 times (after 3 eager lock-steps) leaves exactly the heightmaps, per-bin
    records, observations, masks and episode accumulators the eager loop leaves (the by-value bpp_masked_act with step = t)."""
    import torch
    import bpp_amd
    size = (10, 10, 10)
    pool = bpp_amd.sequences.cut2_pool(size, 256, seed=0)
    torch.manual_seed(0)
    M = 100 * (1 + rot)
    Wt = (torch.randn(400, M, device="cuda") * 0.05)

    def build():
        env = bpp_amd.BppVecEnv(E, size, enable_rotation=rot, pool=pool)
        return env, env.reset(), env.location_masks

    steps = 40
    env_a, obs_a, mask_a = build()
    for t in range(steps):
        action, _ = bpp_amd.masked_act(obs_a @ Wt, mask_a, seed=7, step=t)
        env_a.step_tensors(action)

    env_b, obs_b, mask_b = build()
    counter = torch.tensor([7, 0], dtype=torch.int64, device="cuda")
    out = (torch.empty((E, 1), dtype=torch.int64, device="cuda"), torch.empty((E, 1), dtype=torch.float32, device="cuda"))

    def one_step():
        bpp_amd.masked_act(obs_b @ Wt, mask_b, counter=counter, out=out)
        env_b.step_tensors(out[0])
        counter[1:].add_(1)

    warm = 3
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            one_step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        one_step()
    for _ in range(steps - warm):          # (the capture itself executes nothing)
        g.replay()
    torch.cuda.synchronize()
    assert int(counter[1].item()) == steps
    for name in ("hmap", "state", "ep_acc"):
        assert torch.equal(getattr(env_a, name), getattr(env_b, name)), name
    assert torch.equal(obs_a, obs_b) and torch.equal(mask_a, mask_b)
    assert float(env_b.ep_acc[:, 3].sum().item()) > 0           # episodes finished inside the replays
