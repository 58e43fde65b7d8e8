"""Training half of the masked policy head (acktr/distributions.py:71-101 through Policy.evaluate_actions,
acktr/model.py:90-96, consumed by acktr/algo/acktr_pipeline.py:45-66): log-probability of the taken action, entropy and
the invalid-action probability mass, forward and backward.  Floating point, so the checker is plain PyTorch float32
autograd of the reference's own formula.  Tolerances: forward |diff| <= max(5e-6, 2.5e-8 M) (log-probabilities),
max(2e-5, 1e-7 M) (entropy: a sum of M products), 2e-6 (bad mass); gradients |diff| <= 2e-6 + 1e-4 |g| (float32 sums in a different order)."""
import numpy as np
import pytest
import torch

from test_masked_act import make_case


def torch_reference(x, m, a, w):
    """Reference formula + autograd.  w = (g_logp [E], g_ent [E], g_bad [E]): weights of the three outputs in the loss."""
    x = torch.from_numpy(x).clone().requires_grad_(True)
    m = torch.from_numpy(m)
    lx = torch.softmax(x - (1.0 - m) * 14.0, dim=-1) + 1e-5           # distributions.py:76-80
    d = torch.distributions.Categorical(probs=lx)                       # FixedCategorical(probs=lx)
    logp = d.log_prob(torch.from_numpy(a))                              # dist.log_probs(action)
    ent = d.entropy()                                                   # dist.entropy()
    bad = (torch.softmax(x, dim=-1) * (1.0 - m)).sum(-1)                # bx, distributions.py:86-88
    loss = (torch.from_numpy(w[0]) * logp + torch.from_numpy(w[1]) * ent + torch.from_numpy(w[2]) * bad).sum()
    loss.backward()
    return logp.detach().numpy(), ent.detach().numpy(), bad.detach().numpy(), x.grad.numpy()


def case(E, M, seed):
    x, m = make_case(E, M, seed)
    rng = np.random.RandomState(seed + 1)
    a = rng.randint(0, M, size=E).astype(np.int64)
    feas = m.argmax(1)                                                  # half of the actions on feasible cells
    a[::2] = np.where(m[::2].any(1), feas[::2], a[::2])
    w = tuple(rng.randn(E).astype(np.float32) for _ in range(3))
    return x, m, a, w


def check(fwd, bwd, E, M, seed):
    x, m, a, w = case(E, M, seed)
    lp0, h0, b0, g0 = torch_reference(x, m, a, w)
    lp, h, b = fwd(x, m, a)
    np.testing.assert_allclose(lp, lp0, rtol=0, atol=max(5e-6, 2.5e-8 * M))       # sequential sums of M terms in the oracle
    np.testing.assert_allclose(h, h0, rtol=0, atol=max(2e-5, 1e-7 * M))
    np.testing.assert_allclose(b, b0, rtol=0, atol=2e-6)
    g = bwd(x, m, a, *w)
    np.testing.assert_allclose(g, g0, rtol=1e-4, atol=2e-6)
    # each output on its own (catches a term leaking into another one's gradient)
    for i in range(3):
        wi = tuple(w[j] if j == i else np.zeros(E, np.float32) for j in range(3))
        np.testing.assert_allclose(bwd(x, m, a, *wi), torch_reference(x, m, a, wi)[3], rtol=1e-4, atol=2e-6)


SHAPES = [(300, 100), (130, 200), (50, 400), (64, 8), (20, 800), (5, 37)]


@pytest.mark.parametrize("E,M", SHAPES)
def test_oracle_masked_evaluate_matches_torch_autograd(oracle, E, M):
    check(oracle.masked_evaluate, oracle.masked_evaluate_backward, E, M, seed=E + M)


@pytest.mark.parametrize("E,M", [(70, 100), (9, 200), (5, 37)])
def test_emulated_masked_evaluate_matches_torch_autograd(emu, E, M):
    """The product kernels (wave-level reductions included), compiled by g++ against the SIMT emulator."""
    check(emu.masked_evaluate, emu.masked_evaluate_backward, E, M, seed=E + M)


def test_oracle_masked_evaluate_against_the_reference_policy(oracle):
    """Build container only: the REFERENCE's Policy.evaluate_actions on oracle-produced observations / masks."""
    from oracle import ref_shims
    if not ref_shims.available():
        pytest.skip("reference tree not present")
    import types
    import bpp_amd
    ref_shims.install()
    from acktr.model import Policy
    size, E = (10, 10, 10), 48
    args = types.SimpleNamespace(channel=4, container_size=size, pallet_size=10, enable_rotation=False)
    torch.manual_seed(0)
    policy = Policy((400,), bpp_amd.Discrete(100), base_kwargs={"recurrent": False, "hidden_size": 256, "args": args})
    env = oracle.OracleEnv(bpp_amd.sequences.cut2_pool(size, 16, seed=0), size, False, E)
    obs, mask = env.reset()
    for t in range(5):
        o = env.step(oracle.sample_feasible(mask, 3, t))
        obs, mask = o["obs"], o["mask"]
    action = oracle.sample_feasible(mask, 4, 0)
    ot, mt, at = torch.from_numpy(obs), torch.from_numpy(mask), torch.from_numpy(action).unsqueeze(1)
    _, features, _, _ = policy.base(ot, None, None)
    logits = policy.dist.linear(features)
    logits.retain_grad()
    _, logp, ent, _, bad_prob, _ = policy.evaluate_actions(ot, None, None, at, mt)
    lp, h, b = oracle.masked_evaluate(logits.detach().numpy(), mask, action)
    np.testing.assert_allclose(lp, logp.detach().numpy()[:, 0], atol=5e-6)
    np.testing.assert_allclose(h.mean(), ent.item(), atol=2e-5)
    np.testing.assert_allclose(b.sum() / b.size / 100, bad_prob.mean().item(), atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("E,M", SHAPES + [(4096, 100)])
def test_gpu_masked_evaluate_matches_torch_autograd_and_oracle(oracle, E, M):
    import bpp_amd
    from bpp_amd.masks import _MaskedEvaluate

    def fwd(x, m, a):
        out = _MaskedEvaluate.apply(torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda(), torch.from_numpy(a).cuda())
        return tuple(t.cpu().numpy() for t in out)

    def bwd(x, m, a, g0, g1, g2):
        xt = torch.from_numpy(x).cuda().requires_grad_(True)
        lp, h, b = _MaskedEvaluate.apply(xt, torch.from_numpy(m).cuda(), torch.from_numpy(a).cuda())
        (torch.from_numpy(g0).cuda() * lp + torch.from_numpy(g1).cuda() * h + torch.from_numpy(g2).cuda() * b).sum().backward()
        return xt.grad.cpu().numpy()

    check(fwd, bwd, E, M, seed=E + M)
    x, m, a, w = case(E, M, E + M)
    for got, want, tol in zip(fwd(x, m, a), oracle.masked_evaluate(x, m, a), (max(5e-6, 2.5e-8 * M), max(2e-5, 1e-7 * M), 2e-6)):
        np.testing.assert_allclose(got, want, rtol=0, atol=tol)
    np.testing.assert_allclose(bwd(x, m, a, *w), oracle.masked_evaluate_backward(x, m, a, *w), rtol=1e-4, atol=2e-6)
    # the public wrapper: the three scalars acktr_pipeline.py:45-66 consumes, differentiable end to end
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    logp, ent, prob_loss = bpp_amd.masked_evaluate(xt, torch.from_numpy(m).cuda(), torch.from_numpy(a).cuda().unsqueeze(1))
    assert logp.shape == (E, 1) and ent.dim() == 0 and prob_loss.dim() == 0
    adv = torch.from_numpy(w[0]).cuda().unsqueeze(1)
    (-(adv * logp).mean() - 0.01 * ent + 0.1 * prob_loss).backward()
    xr = torch.from_numpy(x).clone().requires_grad_(True)
    mr = torch.from_numpy(m)
    d = torch.distributions.Categorical(probs=torch.softmax(xr - (1 - mr) * 14, -1) + 1e-5)
    bx = torch.softmax(xr, -1) * (1 - mr)
    (-(torch.from_numpy(w[0]) * d.log_prob(torch.from_numpy(a))).mean() - 0.01 * d.entropy().mean() + 0.1 * bx.mean()).backward()
    np.testing.assert_allclose(xt.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-6 / E + 1e-9)
