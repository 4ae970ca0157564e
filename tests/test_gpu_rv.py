"""GPU parity: the fused radial-velocity op (exo_radial_velocity_*) against the oracle, and
KeplerianOrbit.get_radial_velocity through it against the composed formulas
(/root/reference/src/exoplanet/orbits/keplerian.py:633-677)."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P

pytestmark = pytest.mark.gpu


def T(a, dev, grad=False):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev).requires_grad_(grad)


def npy(x):
    return x.detach().cpu().numpy()


def random_params(rng, D, Pn):
    period = 10 ** rng.uniform(0, 2.5, (D, Pn))
    e = np.where(rng.uniform(size=(D, Pn)) < 0.2, 0.0, rng.uniform(0, 0.9, (D, Pn)))
    w = rng.uniform(-np.pi, np.pi, (D, Pn))
    return np.stack([2 * np.pi / period, rng.uniform(0, 50, (D, Pn)), e, np.cos(w), np.sin(w),
                     10 ** rng.uniform(-1, 2.5, (D, Pn))], axis=-1)


@pytest.mark.parametrize("D,Pn,N", [(1, 1, 1), (1, 2, 300), (7, 3, 1000), (64, 1, 257)])
def test_rv_op_parity(dev, D, Pn, N):
    from exoplanet_amd import ops

    rng = np.random.default_rng(D + 10 * Pn)
    params = random_params(rng, D, Pn)
    t = np.sort(rng.uniform(0, 400, N))
    g = rng.normal(size=(D, N, Pn))
    want, want_g = P.radial_velocity_vjp(t, params, g)
    pt = T(params, dev, True)
    rv = ops.radial_velocity(T(t, dev), pt)
    np.testing.assert_allclose(npy(rv), want, rtol=0, atol=1e-12 * np.abs(want).max())
    (gp,) = torch.autograd.grad((rv * T(g, dev)).sum(), pt)
    scale = np.abs(want_g).max(axis=(0, 1), keepdims=True) + 1e-300
    assert (np.abs(npy(gp) - want_g) / scale).max() < 1e-10
    # bit-reproducible reverse pass
    (gp2,) = torch.autograd.grad((ops.radial_velocity(T(t, dev), pt) * T(g, dev)).sum(), pt)
    assert torch.equal(gp, gp2)


def test_rv_op_edge_cases(dev):
    from exoplanet_amd import ops

    params = random_params(np.random.default_rng(0), 2, 2)
    params[1, 0, P.RV_ECC] = 1.2                                   # outside [0, 1): NaN for that planet only
    t = T(np.linspace(0, 10, 50), dev)
    rv = npy(ops.radial_velocity(t, T(params, dev)))
    assert np.isnan(rv[1, :, 0]).all() and np.isfinite(rv[0]).all() and np.isfinite(rv[1, :, 1]).all()
    assert ops.radial_velocity(T(np.zeros(0), dev), T(params, dev)).shape == (2, 0, 2)
    with pytest.raises(ValueError):
        ops.radial_velocity(t, T(params[..., :5], dev))
    with pytest.raises(RuntimeError):
        ops.radial_velocity(t.cpu(), T(params, dev))


ORBIT = dict(m_star=1.3, r_star=1.0, t0=np.array([0.5, 3.1]), period=np.array([100.0, 37.3]), ecc=np.array([0.1, 0.45]),
             omega=np.array([0.5, -2.0]), incl=np.array([0.25 * np.pi, 1.3]), m_planet=np.array([0.1, 0.02]))


def composed_rv(orbit, t, K=None):
    """the reference's two formulas, op by op (ops.kepler + torch)"""
    from exoplanet_amd.orbits.constants import m_per_s_per_Rsun_per_day

    if K is None:
        return -m_per_s_per_Rsun_per_day * orbit.get_star_velocity(t)[2]
    sinf, cosf = orbit._get_true_anomaly(t)
    if orbit.ecc is None:
        return K * cosf
    return K * (orbit.cos_omega * cosf - orbit.sin_omega * sinf + orbit.ecc * orbit.cos_omega)


@pytest.mark.parametrize("with_K", [False, True])
@pytest.mark.parametrize("circular", [False, True])
def test_orbit_rv_equals_composed(dev, with_K, circular):
    import exoplanet_amd as xo

    kw = {k: (T(v, dev) if isinstance(v, np.ndarray) else v) for k, v in ORBIT.items()}
    if circular:
        kw.pop("ecc"), kw.pop("omega")
    t = T(np.linspace(0, 100, 500), dev)
    K = T(np.array([3.0, 11.0]), dev) if with_K else None
    grads = []
    for route in ("fused", "composed"):
        leaves = {k: v.clone().requires_grad_(True) for k, v in kw.items() if isinstance(v, torch.Tensor)}
        orbit = xo.KeplerianOrbit(**{**kw, **leaves})
        rv = orbit.get_radial_velocity(t, K=K) if route == "fused" else composed_rv(orbit, t, K)
        assert tuple(rv.shape) == (500, 2)
        w = torch.linspace(0.5, 1.5, 1000, dtype=torch.float64, device=rv.device).reshape(500, 2)
        grads.append((rv.detach(), torch.autograd.grad((rv * w).sum(), list(leaves.values()), allow_unused=True)))
    (rv_f, g_f), (rv_c, g_c) = grads
    np.testing.assert_allclose(npy(rv_f), npy(rv_c), rtol=0, atol=1e-12 * float(rv_c.abs().max()))
    for a, b in zip(g_f, g_c):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0
            continue
        np.testing.assert_allclose(npy(a), npy(b), rtol=1e-9, atol=1e-10 * float(b.abs().max()))


def test_orbit_rv_draws(dev):
    """parameters with a leading draw dimension: every draw equals its own evaluation"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(5)
    D = 4
    kw = dict(period=T(10 ** rng.uniform(0.5, 2, (D, 2)), dev), t0=T(rng.uniform(0, 5, (D, 2)), dev),
              ecc=T(rng.uniform(0, 0.6, (D, 2)), dev), omega=T(rng.uniform(-3, 3, (D, 2)), dev), b=T(rng.uniform(0, 0.5, (D, 2)), dev))
    K = T(rng.uniform(1, 20, (D, 2)), dev)
    t = T(np.linspace(0, 60, 200), dev)
    rv = xo.KeplerianOrbit(**kw).get_radial_velocity(t, K=K)
    assert tuple(rv.shape) == (D, 200, 2)
    for d in range(D):
        one = xo.KeplerianOrbit(**{k: v[d] for k, v in kw.items()}).get_radial_velocity(t, K=K[d])
        assert torch.equal(rv[d], one)


def test_orbit_rv_standard_parameterisation_uses_the_packing_kernel(dev):
    """(period, t0, b, ecc, omega) + K: the RV records come from the packing kernel, the orbit's
    attribute algebra never runs; same values and gradients as the composed formulas"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(6)
    base = dict(period=np.array([12.3, 41.0]), t0=np.array([1.0, 7.5]), ecc=np.array([0.2, 0.55]),
                omega=np.array([0.7, -1.9]), b=np.array([0.1, 0.4]))
    t = T(np.sort(rng.uniform(0, 120, 400)), dev)
    out = []
    for route in ("fused", "composed"):
        leaves = {k: T(v, dev, True) for k, v in base.items() if k != "b"}
        K = T(np.array([3.0, 11.0]), dev, True)
        orbit = xo.KeplerianOrbit(b=T(base["b"], dev), **leaves)
        if route == "fused":
            rv = orbit.get_radial_velocity(t, K=K)
            assert not orbit._ready
        else:
            rv = composed_rv(orbit, t, K)
        w = torch.linspace(0.5, 1.5, 800, dtype=torch.float64, device=rv.device).reshape(400, 2)
        out.append((rv.detach(), torch.autograd.grad((rv * w).sum(), list(leaves.values()) + [K])))
    np.testing.assert_allclose(npy(out[0][0]), npy(out[1][0]), rtol=0, atol=1e-12 * float(out[1][0].abs().max()))
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_allclose(npy(a), npy(b), rtol=1e-9, atol=1e-10 * float(b.abs().max()))
