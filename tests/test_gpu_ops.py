"""GPU parity: elementwise ops through the C ABI vs the oracle (numpy port)."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P

pytestmark = pytest.mark.gpu


def T(a, dev, grad=False):
    x = torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
    return x.requires_grad_(grad)


def test_kepler_parity(dev):
    from exoplanet_amd import ops

    rng = np.random.default_rng(11)
    e = np.concatenate([rng.uniform(0, 1, 200000), 1 - 10 ** rng.uniform(-10, 0, 50000), np.zeros(6000)])
    M = np.concatenate([rng.uniform(-50, 50, 200000), 10 ** rng.uniform(-14, 0.49, 50000) * rng.choice([-1, 1], 50000),
                        rng.uniform(-500, 500, 6000)])
    s, c = ops.kepler(T(M, dev), T(e, dev))
    S, C = P.kepler(M, e)
    # conditioning w.r.t. the rounding of M itself: eps |M_red| df/dM
    cond = 1 + (1 + e * C) ** 2 / (1 - e * e) ** 1.5 * np.abs(np.remainder(M + np.pi, 2 * np.pi) - np.pi)
    err = np.maximum(np.abs(s.cpu().numpy() - S), np.abs(c.cpu().numpy() - C))
    assert np.all(err <= 8 * 2.3e-16 * cond)


def test_kepler_invalid_ecc_is_nan(dev):
    from exoplanet_amd import ops

    s, c = ops.kepler(T([0.3, 0.3, 0.3], dev), T([-0.1, 1.0, 0.5], dev))
    s = s.cpu().numpy()
    assert np.isnan(s[0]) and np.isnan(s[1]) and np.isfinite(s[2])


def test_kepler_grad(dev):
    from exoplanet_amd import ops

    rng = np.random.default_rng(3)
    M = T(rng.uniform(-10, 10, 40), dev, True)
    e = T(rng.uniform(0, 0.9, 40), dev, True)
    assert torch.autograd.gradcheck(lambda a, b: ops.kepler(a, b), (M, e), eps=1e-7, atol=1e-6, rtol=1e-6)


def test_quad_sv_parity(dev):
    from exoplanet_amd import ops

    rng = np.random.default_rng(5)
    r = np.concatenate([rng.uniform(0.001, 1.5, 100000), 10 ** rng.uniform(-3, 1.2, 50000), np.full(20000, 0.1)])
    b = np.concatenate([rng.uniform(0, 3, 100000),
                        rng.uniform(0, 1, 50000) * (1 + 10 ** rng.uniform(-3, 1.2, 50000)) * 1.02,
                        rng.uniform(-1.5, 1.5, 20000)])
    rs = rng.uniform(0.01, 1.2, 10000)
    eps = 10 ** rng.uniform(-14, -3, 10000) * rng.choice([-1, 1], 10000)
    b = np.concatenate([b, rs + eps, np.abs(1 - rs) + eps, 1 + rs + eps, np.abs(eps)])
    r = np.concatenate([r, rs, rs, rs, rs])
    bt, rt = T(b, dev, True), T(r, dev, True)
    s = ops.quad_solution_vector(bt, rt)
    S, DB, DR = P.quad_solution_vector(b, r)
    assert np.abs(s.detach().cpu().numpy() - S).max() < 2e-14
    w = rng.normal(size=S.shape)
    (s * T(w, dev)).sum().backward()
    assert np.abs(bt.grad.cpu().numpy() - (w * DB).sum(-1)).max() < 2e-13
    assert np.abs(rt.grad.cpu().numpy() - (w * DR).sum(-1)).max() < 2e-13


def test_quad_sv_edge_cases(dev):
    from exoplanet_amd import ops

    b = np.array([0.0, 0.0, 0.5, 5.0, 0.2, np.nan, 0.3, 0.0])
    r = np.array([0.1, 1.0, 0.5, 0.1, 2.0, 0.1, 0.0, 0.0])
    s = ops.quad_solution_vector(T(b, dev), T(r, dev)).cpu().numpy()
    S, _, _ = P.quad_solution_vector(b, r)
    np.testing.assert_allclose(s, S, rtol=0, atol=2e-15, equal_nan=True)
    empty = ops.quad_solution_vector(T([], dev), T([], dev))
    assert empty.shape == (0, 3)


def test_contact_points_parity(dev):
    from exoplanet_amd import ops

    rng = np.random.default_rng(9)
    n = 300
    a = rng.uniform(3, 60, n); e = rng.uniform(0, 0.9, n); w = rng.uniform(-np.pi, np.pi, n)
    bimp = rng.uniform(0, 1.4, n)
    incl_factor = (1 + e * np.sin(w)) / (1 - e * e)
    cosi = incl_factor * bimp / a
    sini = np.sqrt(np.clip(1 - cosi ** 2, 0, None))
    L = 1 + rng.uniform(0.01, 0.2, n)
    Ml, Mr, fl = ops.contact_points(*[T(x, dev) for x in (a, e, np.cos(w), np.sin(w), cosi, sini, L)])
    ml, mr, f0 = P.contact_points(a, e, np.cos(w), np.sin(w), cosi, sini, L)
    assert np.array_equal(fl.cpu().numpy(), f0)
    ok = f0 == 0
    assert ok.sum() > 100 and (~ok).sum() > 5
    np.testing.assert_allclose(Ml.cpu().numpy()[ok], ml[ok], rtol=0, atol=1e-12)
    np.testing.assert_allclose(Mr.cpu().numpy()[ok], mr[ok], rtol=0, atol=1e-12)


def test_host_tensor_is_rejected(dev):
    from exoplanet_amd import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.kepler(torch.zeros(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64))
