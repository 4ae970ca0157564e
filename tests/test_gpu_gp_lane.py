"""GPU: the one-lane-per-(draw, chunk) celerite kernels with a checkpointed factorisation
(J <= 2; exo_celerite_core.hpp, the same code tests/test_gp_host.py runs on the CPU), the per-draw
pair kinds, and the statelessness of the library -- through the C ABI, against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import c_port as C
from oracle import numpy_port as P
from test_gpu_gp import T

pytestmark = pytest.mark.gpu


def sho(sigma, rho, Q):
    return P.sho_coefficients(*P.sho_from_sigma_rho(sigma, rho, Q), Q)


def test_c3_shape_lane_path_vs_c_port(dev):
    """C3's kernel and sampling at 30 000 cadences, default plan (one-lane kernels): log-likelihood
    and every gradient against the C port's sequential recurrences, and against this library's own
    sequential kernels (n_chunks = 1)"""
    from exoplanet_amd.gp import celerite_loglike

    rng = np.random.default_rng(21)
    n, D = 30_000, 6
    t = np.arange(n) * (2.0 / 1440.0)
    co = sho(1e-3, 5.0, 1 / np.sqrt(2))
    cplx = np.repeat(np.stack(co[2:], -1)[None], D, 0)
    cplx[:, :, 0] *= 1 + 0.05 * rng.normal(size=(D, 1))
    cplx[:, :, 1] = cplx[:, :, 0] * (co[3][0] / co[2][0])
    y = 5e-4 * rng.normal(size=(D, n))
    diag = np.full((1, n), 2.5e-7)
    real = np.zeros((D, 0, 2))
    w = np.linspace(0.5, 1.5, D)
    out = {}
    for n_chunks in (None, 1, 37):
        yt, ct, dt = T(y, dev, True), T(cplx, dev, True), T(diag, dev, True)
        ll = celerite_loglike(T(t, dev), yt, dt, T(real, dev), ct, n_chunks=n_chunks)
        (ll * T(w, dev)).sum().backward()
        out[n_chunks] = [x.detach().cpu().numpy() for x in (ll, yt.grad, dt.grad, ct.grad)]
    for key in (None, 37):
        assert not np.array_equal(out[key][0], out[1][0])          # another summation order: the chunked kernels ran
        np.testing.assert_allclose(out[key][0], out[1][0], rtol=1e-12)
        for a, b in zip(out[key][1:], out[1][1:]):
            assert np.abs(a - b).max() <= 2e-9 * np.abs(b).max()
    z = np.zeros(0)
    for d in (0, D - 1):
        want, gw = C.celerite(t, y[d], diag[0], (z, z, cplx[d, :, 0], cplx[d, :, 1], cplx[d, :, 2], cplx[d, :, 3]), grad=True)
        assert abs(out[None][0][d] - want) <= 1e-12 * abs(want)
        np.testing.assert_allclose(out[None][1][d], w[d] * gw["y"], rtol=1e-7, atol=1e-9 * np.abs(gw["y"]).max())
        for q, key in enumerate(("ac", "bc", "cc", "dc")):
            np.testing.assert_allclose(out[None][3][d, 0, q], w[d] * gw[key][0], rtol=2e-6)


@pytest.mark.parametrize("n", [64, 203, 4097])
def test_lane_path_ragged_lengths_and_gaps(dev, n):
    """series lengths that are no multiple of the checkpoint block or the chunk length, unevenly
    sampled with a gap, per-draw diag, J = 1 and J = 2 (real + real, one complex)"""
    from exoplanet_amd.gp import celerite_loglike

    rng = np.random.default_rng(n)
    D = 5
    t = np.sort(rng.uniform(0, 0.2 * n, n))
    t[n // 3:] += 7.0
    y = rng.normal(size=(D, n))
    diag = rng.uniform(0.05, 0.3, (D, n))
    for n_real, n_complex in ((1, 0), (2, 0), (0, 1)):
        real = np.stack([rng.uniform(0.3, 1.5, (D, n_real)), rng.uniform(0.05, 2.0, (D, n_real))], -1)
        ac, cc, dc = rng.uniform(0.3, 1.5, (D, n_complex)), rng.uniform(0.05, 1.0, (D, n_complex)), rng.uniform(0.3, 4, (D, n_complex))
        cplx = np.stack([ac, rng.uniform(-0.9, 0.9, (D, n_complex)) * ac * cc / dc, cc, dc], -1)
        res = {}
        for n_chunks in (None, 1):
            yt, dt = T(y, dev, True), T(diag, dev, True)
            rt, ct = T(real, dev, True), T(cplx, dev, True)
            ll = celerite_loglike(T(t, dev), yt, dt, rt, ct, n_chunks=n_chunks)
            ll.sum().backward()
            res[n_chunks] = [x.detach().cpu().numpy() for x in (ll, yt.grad, dt.grad, rt.grad, ct.grad)]
        np.testing.assert_allclose(res[None][0], res[1][0], rtol=1e-12)
        for a, b in zip(res[None][1:], res[1][1:]):
            if b.size:
                assert np.abs(a - b).max() <= 5e-9 * np.abs(b).max()
        if n <= 256:
            for d in range(D):
                co = (real[d, :, 0], real[d, :, 1], cplx[d, :, 0], cplx[d, :, 1], cplx[d, :, 2], cplx[d, :, 3])
                want, gw = P.gp_loglike_dense(t, y[d], diag[d], co)
                assert abs(res[None][0][d] - want) <= 1e-11 * abs(want)
                np.testing.assert_allclose(res[None][1][d], gw["y"], rtol=1e-8, atol=1e-10)
                np.testing.assert_allclose(res[None][2][d], gw["diag"], rtol=1e-8, atol=1e-10)


def test_batch_straddling_q_half(dev):
    """SHO draws on both sides of Q = 1/2 in ONE batch (pair kinds decided on the device): every
    draw against the oracle evaluated with its own regime's coefficients; gradients flow to
    (sigma, rho, Q) of every draw"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(22)
    n = 3000
    t = np.sort(rng.uniform(0, 60, n))
    y = 0.3 * rng.normal(size=n)
    Qs = np.array([0.3, 3.0, 0.45, 0.7071, 0.2, 1.5])
    sig, rho = np.full(Qs.size, 0.4), np.linspace(2.0, 6.0, Qs.size)
    sigma_t, rho_t, Q_t = T(sig, dev, True), T(rho, dev, True), T(Qs, dev, True)
    gp = xo.gp.GaussianProcess(xo.gp.terms.SHOTerm(sigma=sigma_t, rho=rho_t, Q=Q_t), t=T(t, dev), yerr=0.1)
    ll = gp.log_likelihood(T(y, dev))
    ll.sum().backward()
    got = ll.detach().cpu().numpy()
    for d, Q in enumerate(Qs):
        want = P.celerite_loglike(t, y, np.full(n, 0.01), sho(sig[d], rho[d], Q))
        assert abs(got[d] - want) <= 1e-10 * abs(want), (d, Q)
    for g in (sigma_t.grad, rho_t.grad, Q_t.grad):
        assert bool(torch.isfinite(g).all()) and float(g.abs().min()) > 0
    # central differences of the oracle in Q for one draw of each regime
    for d in (0, 1):
        h = 1e-6
        f = lambda q: P.celerite_loglike(t, y, np.full(n, 0.01), sho(sig[d], rho[d], q))  # noqa: E731
        fd = (f(Qs[d] + h) - f(Qs[d] - h)) / (2 * h)
        assert abs(float(Q_t.grad[d]) - fd) <= 2e-5 * abs(fd) + 1e-6


def test_c3_step_is_capturable_and_matches_eager(dev):
    """light curve -> SHO-term GP log-likelihood -> gradients of every leaf, captured ONCE as a
    hipGraph (terms and the GaussianProcess are built inside the captured function: nothing
    synchronises with the host) and replayed with new parameter values"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(23)
    n, D = 6000, 8
    t = torch.arange(n, dtype=torch.float64, device=dev) * (2.0 / 1440.0)
    yobs = 5e-4 * torch.randn(n, dtype=torch.float64, device=dev)
    mk = lambda v: torch.tensor(v * (1 + 1e-3 * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev,  # noqa: E731
                                requires_grad=True)
    leaves = [mk(3.5), mk(1.0), mk(0.3), mk(0.3), mk(1.1), mk(0.1)]
    hyper = [torch.full((D,), v, dtype=torch.float64, device=dev, requires_grad=True) for v in (1e-3, 5.0, 0.7)]
    hyper[2].data[::2] = 0.4       # half of the draws over-damped

    def step(period, t0, b, ecc, omega, r, sigma, rho, Q):
        orbit = xo.KeplerianOrbit(period=period, t0=t0, b=b, ecc=ecc, omega=omega)
        lc = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=r, t=t)
        gp = xo.gp.GaussianProcess(xo.gp.terms.SHOTerm(sigma=sigma, rho=rho, Q=Q), t=t, yerr=5e-4, mean=lc.sum(-1))
        ll = gp.log_likelihood(yobs)
        return (ll.detach(),) + torch.autograd.grad(ll.sum(), (period, t0, b, ecc, omega, r, sigma, rho, Q))

    eager = [x.clone() for x in step(*leaves, *hyper)]
    g = xo.GraphedStep(step, *leaves, *hyper)
    replay = [x.clone() for x in g()]
    for a, b in zip(eager, replay):
        assert torch.equal(a, b)
    # new values through the static inputs
    new = [x.detach() * (1 + 1e-4) for x in leaves] + [x.detach().clone() for x in hyper]
    replay2 = [x.clone() for x in g(*new)]
    with torch.enable_grad():
        fresh = step(*[x.clone().requires_grad_(True) for x in new])
    for a, b in zip(fresh, replay2):
        assert torch.allclose(a, b, rtol=1e-12, atol=0)


def test_library_keeps_no_state_between_calls(dev):
    """> 4096 forward calls before the first backward, with EXO_GP_CHUNKS changing in between: each
    backward reads its own forward's plan from the arguments the autograd context carries"""
    from exoplanet_amd.gp import celerite_loglike

    rng = np.random.default_rng(24)
    n, D = 256, 2
    t = np.sort(rng.uniform(0, 30, n))
    co = sho(0.4, 3.0, 2.0)
    cplx = T(np.repeat(np.stack(co[2:], -1)[None], D, 0), dev)
    real = T(np.zeros((D, 0, 2)), dev)
    diag = T(np.full((1, n), 0.05), dev)
    tt = T(t, dev)
    ys = [T(rng.normal(size=(D, n)), dev, True) for _ in range(2)]
    old = os.environ.get("EXO_GP_CHUNKS")
    try:
        os.environ["EXO_GP_CHUNKS"] = "5"
        first = celerite_loglike(tt, ys[0], diag, real, cplx)
        os.environ["EXO_GP_CHUNKS"] = "3"
        for _ in range(4200):
            celerite_loglike(tt, ys[1], diag, real, cplx)
        os.environ["EXO_GP_CHUNKS"] = "7"
        first.sum().backward()
        got = ys[0].grad.clone()
    finally:
        if old is None:
            os.environ.pop("EXO_GP_CHUNKS", None)
        else:
            os.environ["EXO_GP_CHUNKS"] = old
    yref = ys[0].detach().clone().requires_grad_(True)
    ref = celerite_loglike(tt, yref, diag, real, cplx, n_chunks=1)
    ref.sum().backward()
    assert torch.allclose(first.detach(), ref.detach(), rtol=1e-12)
    assert float((got - yref.grad).abs().max()) <= 2e-9 * float(yref.grad.abs().max())
