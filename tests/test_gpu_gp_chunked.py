"""GPU: the time-parallel celerite path (chunk elements + per-draw scan over chunks + ordinary
recurrences per chunk) against the sequential kernels and the dense oracle.

EXO_GP_CHUNKS (read by the Python wrapper, handed to the library as the n_chunks argument of every
call of a pair) forces the number of chunks; `chunks(0)` below = sequential recurrences (n_chunks = 1)."""
import os

import numpy as np
import pytest
import torch

from oracle import numpy_port as P
from test_gpu_gp import KERNELS, T, _pack

pytestmark = pytest.mark.gpu


class chunks:
    def __init__(self, c):
        self.c = c

    def __enter__(self):
        self.old = os.environ.get("EXO_GP_CHUNKS")
        if self.c is None:
            os.environ.pop("EXO_GP_CHUNKS", None)
        else:
            os.environ["EXO_GP_CHUNKS"] = str(self.c if self.c else 1)   # 1 = sequential recurrences

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("EXO_GP_CHUNKS", None)
        else:
            os.environ["EXO_GP_CHUNKS"] = self.old


def value_and_grads(dev, t, y, diag, cr, cc):
    from exoplanet_amd.gp import celerite_loglike

    tt, yt, dt = T(t, dev), T(y, dev, True), T(diag, dev, True)
    crt, cct = T(cr, dev, True), T(cc, dev, True)
    ll = celerite_loglike(tt, yt, dt, crt, cct)
    w = torch.linspace(0.5, 1.5, ll.numel(), dtype=torch.float64, device=ll.device)   # non-trivial cotangent
    (ll * w).sum().backward()
    return [x.detach().cpu().numpy() for x in (ll, yt.grad, dt.grad, crt.grad, cct.grad)]


def batch(rng, names, D):
    """D draws: every draw gets the same term structure (KERNELS[name]) with jittered coefficients"""
    co0 = KERNELS[names]()
    cr0, cc0 = _pack(co0)
    cr = cr0 * (1 + 0.05 * rng.normal(size=(D,) + cr0.shape[1:]))
    cc = cc0 * (1 + 0.02 * rng.normal(size=(D,) + cc0.shape[1:]))
    if cc.shape[1]:
        # keep every complex term a valid kernel after the jitter: |b d| <= a c
        a, b, c, d = (cc[..., k] for k in range(4))
        cc[..., 1] = np.sign(b) * np.minimum(np.abs(b), 0.999 * a * c / np.abs(d))
    return cr, cc


@pytest.mark.parametrize("name", ["sho_q3", "real1", "mixed_j5", "three_sho_j6"])
@pytest.mark.parametrize("C", [2, 7, 33])
def test_chunked_equals_sequential(dev, name, C):
    rng = np.random.default_rng(11)
    N, D = 1500, 9
    t = np.sort(rng.uniform(0, 60, N))
    y = 0.4 * rng.normal(size=(D, N))
    diag = 0.05 + 0.05 * rng.uniform(size=(D, N))          # per-draw measurement variance
    cr, cc = batch(rng, name, D)
    with chunks(0):
        want = value_and_grads(dev, t, y, diag, cr, cc)
    with chunks(C):
        got = value_and_grads(dev, t, y, diag, cr, cc)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-12)
    for g, w in zip(got[1:], want[1:]):
        if w.size:
            assert np.abs(g - w).max() / (np.abs(w).max() + 1e-300) < 2e-9


def test_chunked_vs_dense_oracle_default_plan(dev):
    """the library's own chunk plan, a ragged last chunk, against the dense definition"""
    rng = np.random.default_rng(12)
    N = 777
    t = np.sort(rng.uniform(0, 30, N))
    y = 0.5 * rng.normal(size=N)
    diag = 0.1 + 0.05 * rng.uniform(size=N)
    co = KERNELS["mixed_j5"]()
    want, g = P.gp_loglike_dense(t, y, diag, co)
    cr, cc = _pack(co)
    with chunks(None):
        from exoplanet_amd.gp import celerite_loglike
        tt, yt, dt = T(t, dev), T(y[None], dev, True), T(diag[None], dev, True)
        crt, cct = T(cr, dev, True), T(cc, dev, True)
        ll = celerite_loglike(tt, yt, dt, crt, cct)
        ll.sum().backward()
    assert abs(ll.item() - want) < 1e-10 * abs(want)
    np.testing.assert_allclose(yt.grad.cpu().numpy()[0], g["y"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(dt.grad.cpu().numpy()[0], g["diag"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(crt.grad.cpu().numpy()[0][:, 0], g["ar"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(cct.grad.cpu().numpy()[0][:, 3], g["dc"], rtol=1e-7, atol=1e-8)


def test_draws_outside_the_filter_form_fall_back(dev):
    """A real term with a negative amplitude (valid only as part of the sum) has no positive state
    covariance: such draws are flagged on the device and redone by the sequential kernels, next to
    draws of the same batch that take the time-parallel path."""
    rng = np.random.default_rng(13)
    N, D = 900, 70                      # more than one wave of draws
    t = np.sort(rng.uniform(0, 50, N))
    y = 0.3 * rng.normal(size=(D, N))
    diag = np.full((1, N), 0.02)
    cr = np.zeros((D, 2, 2))
    cc = np.zeros((D, 0, 4))
    neg = rng.uniform(size=D) < 0.3
    assert neg.any() and (~neg).any()
    for d in range(D):
        s = 0.5 * (1 + 0.1 * rng.normal())
        cr[d, 0] = [s, 0.5]
        cr[d, 1] = [(-0.1 if neg[d] else 0.3) * s, 2.0]    # k = s e^{-0.5 tau} - 0.1 s e^{-2 tau} is still a kernel
    with chunks(0):
        want = value_and_grads(dev, t, y, diag, cr, cc)
    with chunks(None):
        got = value_and_grads(dev, t, y, diag, cr, cc)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-12)
    for g, w in zip(got[1:4], want[1:4]):
        assert np.abs(g - w).max() / (np.abs(w).max() + 1e-300) < 2e-9
    # the flagged draws were computed by the very same kernels: bit-identical
    np.testing.assert_array_equal(got[0][neg], want[0][neg])
    # and one of each kind against the dense definition
    for d in (int(np.flatnonzero(neg)[0]), int(np.flatnonzero(~neg)[0])):
        co = (cr[d, :, 0], cr[d, :, 1]) + (np.zeros(0),) * 4
        ref, _ = P.gp_loglike_dense(t, y[d], diag[0], co)
        assert abs(got[0][d] - ref) < 1e-10 * abs(ref)


def test_not_positive_definite_is_minus_inf_on_both_paths(dev):
    from exoplanet_amd.gp import celerite_loglike

    N = 600
    t = np.linspace(0, 10, N)
    y = np.zeros((2, N))
    diag = np.full((1, N), -5.0)        # a_n = diag + sum a < 0
    cr = np.tile(np.array([[[0.3, 0.2]]]), (2, 1, 1))
    cc = np.zeros((2, 0, 4))
    for c in (0, 5):
        with chunks(c):
            ll = celerite_loglike(T(t, dev), T(y, dev), T(diag, dev), T(cr, dev), T(cc, dev))
            # value-only and value + gradient calls
            yt = T(y, dev, True)
            ll2 = celerite_loglike(T(t, dev), yt, T(diag, dev), T(cr, dev), T(cc, dev))
        assert torch.isinf(ll).all() and (ll < 0).all()
        assert torch.isinf(ll2).all() and (ll2 < 0).all()


def test_no_white_noise_falls_back(dev):
    """diag = 0 is a legal celerite model (the kernel matrix itself is positive definite) but has no
    information-form element: flagged by the element kernel, redone sequentially, same numbers"""
    rng = np.random.default_rng(14)
    N, D = 500, 3
    t = np.sort(rng.uniform(0, 20, N))
    y = 0.3 * rng.normal(size=(D, N))
    diag = np.zeros((D, N))
    diag[1] = 0.05                       # one ordinary draw in the batch
    cr, cc = batch(rng, "mixed_j5", D)
    with chunks(0):
        want = value_and_grads(dev, t, y, diag, cr, cc)
    with chunks(None):
        got = value_and_grads(dev, t, y, diag, cr, cc)
    assert np.isfinite(want[0]).all()
    np.testing.assert_array_equal(got[0][[0, 2]], want[0][[0, 2]])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-12)
    for g, w in zip(got[1:], want[1:]):
        if w.size:
            assert np.abs(g - w).max() / (np.abs(w).max() + 1e-300) < 2e-9


def test_conditioning_score_decides_the_path(dev):
    """The element kernel's conditioning score kappa = (1 + max (b/a)^2) sum(a) / min(diag) sends a draw to the
    sequential kernels above 1e7 for J <= 2 (round 3; 1e5 in round 2, and still for wider states): celerite2's Matern-3/2 term
    (b / a = w0 / eps = 58 here) and an SHO term at a signal 1e6 x the noise now stay on the time-parallel path (close
    to the sequential numbers, not identical), a signal 1e10 x the noise does not (identical numbers)."""
    import exoplanet_amd as xo
    from exoplanet_amd.gp import terms

    rng = np.random.default_rng(15)
    N, D = 800, 6
    t = T(np.sort(rng.uniform(0, 30, N)), dev)
    y = T(0.3 * rng.normal(size=(D, N)), dev)
    sig = T(0.5 * (1 + 0.1 * rng.normal(size=D)), dev, True)
    rho = T(3.0 * (1 + 0.1 * rng.normal(size=D)), dev, True)

    def both(make_kernel, yerr):
        out = []
        for c in (0, None):
            with chunks(c):
                gp = xo.gp.GaussianProcess(make_kernel(), t=t, yerr=yerr)
                ll = gp.log_likelihood(y)
                g = torch.autograd.grad(ll.sum(), (sig, rho))
                out.append([x.detach().cpu().numpy() for x in (ll,) + g])
        return out

    def close(seq, par, rtol_ll, rtol_g):
        assert not np.array_equal(seq[0], par[0])              # two different algorithms ...
        np.testing.assert_allclose(par[0], seq[0], rtol=rtol_ll)   # ... one answer
        for a, b in zip(seq[1:], par[1:]):
            err = np.abs(a - b).max() / np.abs(a).max()
            assert err <= rtol_g, (err, rtol_g)

    close(*both(lambda: terms.Matern32Term(sigma=sig, rho=rho), 0.05), 1e-11, 1e-7)      # kappa ~ 3e5
    close(*both(lambda: terms.SHOTerm(sigma=sig, rho=rho, Q=0.8), 0.05), 1e-12, 1e-9)
    # signal 1e6 x the noise AND a series that is nothing like a draw from the process (white noise of variance 0.09
    # against errors of 5e-4).  The coefficient gradients are ~1e11 and both algorithms have them to 3e-10 .. 2e-8 of
    # the long-double values (checked on the host: the case of this test, draw by draw); d / d rho is their sum with
    # four digits of cancellation (3e7), so the two paths meet there at ~5e-5 -- the parameterisation's conditioning,
    # not a path's.  (On draws from the process: test_gpu_golden.py::test_gp_hard_golden[snr1e6], 1e-6 against long double.)
    close(*both(lambda: terms.SHOTerm(sigma=sig, rho=rho, Q=0.8), 5e-4), 1e-9, 3e-4)
    seq, par = both(lambda: terms.SHOTerm(sigma=sig, rho=rho, Q=0.8), 5e-6)              # 1e10 x: flagged
    for a, b in zip(seq, par):
        np.testing.assert_array_equal(a, b)


def test_reverse_pass_follows_the_forward_plan(dev):
    """EXO_GP_CHUNKS changing between the forward and the reverse call (or between the sizing of the
    state buffer and the forward call) must not change what the reverse pass reads"""
    from exoplanet_amd.gp import celerite_loglike

    rng = np.random.default_rng(16)
    N, D = 1200, 5
    t = np.sort(rng.uniform(0, 40, N))
    y = 0.4 * rng.normal(size=(D, N))
    diag = np.full((1, N), 0.05)
    cr, cc = batch(rng, "mixed_j5", D)
    with chunks(0):
        want = value_and_grads(dev, t, y, diag, cr, cc)
    tt, yt, dt = T(t, dev), T(y, dev, True), T(diag, dev)
    crt, cct = T(cr, dev, True), T(cc, dev, True)
    with chunks(9):
        ll = celerite_loglike(tt, yt, dt, crt, cct)
    w = torch.linspace(0.5, 1.5, D, dtype=torch.float64, device=ll.device)
    with chunks(31):
        (ll * w).sum().backward()
    np.testing.assert_allclose(ll.detach().cpu().numpy(), want[0], rtol=1e-12)
    for g, wv in zip((yt.grad, crt.grad, cct.grad), (want[1], want[3], want[4])):
        g = g.cpu().numpy()
        assert np.abs(g - wv).max() / (np.abs(wv).max() + 1e-300) < 2e-9


def test_value_only_calls_take_the_time_parallel_path_too(dev):
    from exoplanet_amd.gp import celerite_loglike

    rng = np.random.default_rng(17)
    N, D = 2000, 4
    t = np.sort(rng.uniform(0, 50, N))
    y = 0.4 * rng.normal(size=(D, N))
    diag = np.full((1, N), 0.05)
    cr, cc = batch(rng, "sho_q3", D)
    with chunks(0):
        want = celerite_loglike(T(t, dev), T(y, dev), T(diag, dev), T(cr, dev), T(cc, dev))
    with chunks(None), torch.no_grad():
        got = celerite_loglike(T(t, dev), T(y, dev), T(diag, dev), T(cr, dev), T(cc, dev))
    assert not torch.equal(got, want)          # a different summation order: the chunked kernels ran
    assert torch.allclose(got, want, rtol=1e-12)


def test_wide_state_takes_the_time_parallel_path(dev):
    """J = 7, 8 (lane-group element kernel): chunked == sequential == dense"""
    rng = np.random.default_rng(18)
    N, D = 900, 3
    t = np.sort(rng.uniform(0, 40, N))
    y = 0.4 * rng.normal(size=(D, N))
    diag = np.full((1, N), 0.05)
    parts = [P.sho_coefficients(*P.sho_from_sigma_rho(s, r, q), q)
             for s, r, q in ((0.4, 20.0, 2.0), (0.3, 10.0, 1.0), (0.2, 2.0, 0.8), (0.25, 5.0, 1.5))]
    for n_real in (1, 0):                                  # J = 7 (3 pairs + 1 real), J = 8 (4 pairs)
        sel = parts[:3] if n_real else parts
        co = tuple(np.concatenate(x) for x in zip(*sel))
        if n_real:
            co = (np.array([0.3]), np.array([0.2])) + co[2:]
        cr0, cc0 = _pack(co)
        cr = np.repeat(cr0, D, 0) * (1 + 0.02 * rng.normal(size=(D,) + cr0.shape[1:]))
        cc = np.repeat(cc0, D, 0)
        cc[..., 0] *= 1 + 0.02 * rng.normal(size=cc[..., 0].shape)
        cc[..., 1] = cc[..., 0] * (cc0[..., 1] / cc0[..., 0])      # keep b / a: SHO terms sit on |b d| = a c
        with chunks(0):
            want = value_and_grads(dev, t, y, diag, cr, cc)
        with chunks(11):
            got = value_and_grads(dev, t, y, diag, cr, cc)
        assert not np.array_equal(got[0], want[0])
        np.testing.assert_allclose(got[0], want[0], rtol=1e-12)
        for g, w in zip(got[1:], want[1:]):
            if w.size:
                assert np.abs(g - w).max() / (np.abs(w).max() + 1e-300) < 2e-9
        co_d = (cr[0, :, 0], cr[0, :, 1], cc[0, :, 0], cc[0, :, 1], cc[0, :, 2], cc[0, :, 3])
        ref, _ = P.gp_loglike_dense(t, y[0], diag[0], co_d)
        assert abs(got[0][0] - ref) < 1e-10 * abs(ref)
        # round 4: the oscillation-rate gradient as a phase flux, phases from the first stamp -- the lane-group kernels too: the
        # series 3000 d away (stamps on a 2^-20 d grid: an exact shift) gives the same gradients
        tq = np.round(t * 2.0 ** 20) / 2.0 ** 20
        with chunks(11):
            a = value_and_grads(dev, tq, y, diag, cr, cc)
            b = value_and_grads(dev, tq + 3000.0, y, diag, cr, cc)
        np.testing.assert_allclose(b[0], a[0], rtol=1e-12)
        for g, w in zip(b[1:], a[1:]):
            if w.size:
                assert np.abs(g - w).max() / (np.abs(w).max() + 1e-300) < 1e-9


@pytest.mark.parametrize("J", [9, 10, 11, 12, 13, 14, 15, 16])
def test_states_wider_than_eight_take_the_time_parallel_path(dev, J):
    """J = 9 .. 16 (round 6): the lane-group element / chunk kernels on a DPP row of sixteen lanes, the scans on blocks of 256
    threads with the matrices in LDS (celerite_tree_wide_kernel): chunked == sequential == dense -- on the default plan and on
    forced ones (an odd number of chunks, a plan with a single-chunk tail), with a batch that does not fill its last wave; the
    same at a time origin 3000 d away"""
    from exoplanet_amd import _lib

    rng = np.random.default_rng(100 + J)
    N, D = 1500, 7
    t = np.sort(rng.uniform(0, 60, N))
    y = 0.4 * rng.normal(size=(D, N))
    diag = np.full((1, N), 0.05)
    sho = [(0.4, 20.0, 2.0), (0.3, 10.0, 1.0), (0.2, 2.0, 0.8), (0.25, 5.0, 1.5), (0.15, 1.1, 4.0), (0.2, 33.0, 0.9),
           (0.1, 0.6, 2.5), (0.12, 7.7, 6.0)]
    n_pair, n_real = J // 2, J % 2
    parts = [P.sho_coefficients(*P.sho_from_sigma_rho(s_, r_, q_), q_) for s_, r_, q_ in sho[:n_pair]]
    co = tuple(np.concatenate(x) for x in zip(*parts))
    if n_real:
        co = (np.array([0.3]), np.array([0.2])) + co[2:]
    cr0, cc0 = _pack(co)
    assert cr0.shape[1] + 2 * cc0.shape[1] == J
    cr = np.repeat(cr0, D, 0) * (1 + 0.02 * rng.normal(size=(D,) + cr0.shape[1:]))
    cc = np.repeat(cc0, D, 0)
    cc[..., 0] *= 1 + 0.02 * rng.normal(size=cc[..., 0].shape)
    cc[..., 1] = cc[..., 0] * (cc0[..., 1] / cc0[..., 0])      # keep b / a: SHO terms sit on |b d| = a c
    assert int(_lib.load().exo_celerite_default_chunks(N, D, cr0.shape[1], cc0.shape[1], 0)) > 1     # the default plan cuts the series
    with chunks(0):
        want = value_and_grads(dev, t, y, diag, cr, cc)
    co_d = (cr[0, :, 0], cr[0, :, 1], cc[0, :, 0], cc[0, :, 1], cc[0, :, 2], cc[0, :, 3])
    ref, _ = P.gp_loglike_dense(t, y[0], diag[0], co_d)
    for C in (None, 11, 32):
        with chunks(C):
            got = value_and_grads(dev, t, y, diag, cr, cc)
        assert not np.array_equal(got[0], want[0])                 # (not the sequential kernels' numbers: the path was taken)
        np.testing.assert_allclose(got[0], want[0], rtol=1e-12)
        for g, w in zip(got[1:], want[1:]):
            if w.size:
                assert np.abs(g - w).max() / (np.abs(w).max() + 1e-300) < 5e-9
        assert abs(got[0][0] - ref) < 1e-10 * abs(ref)
    tq = np.round(t * 2.0 ** 20) / 2.0 ** 20
    with chunks(11):
        a = value_and_grads(dev, tq, y, diag, cr, cc)
        b = value_and_grads(dev, tq + 3000.0, y, diag, cr, cc)
    np.testing.assert_allclose(b[0], a[0], rtol=1e-12)
    for g, w in zip(b[1:], a[1:]):
        if w.size:
            assert np.abs(g - w).max() / (np.abs(w).max() + 1e-300) < 1e-9


def test_wide_state_with_pair_kinds_per_draw(dev):
    """J = 10 from five SHO terms whose first one straddles Q = 1/2 across the batch (pair kinds per draw: a complex term for
    some draws, two real ones for others): the wide time-parallel path against the sequential kernels, value and gradients of the
    term parameters"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(77)
    N, D = 1800, 9
    t = T(np.sort(rng.uniform(0, 50, N)), dev)
    y = T(0.3 * rng.normal(size=N), dev)
    res = {}
    for C in (0, None):
        v = lambda x, s=0.03: torch.tensor(x * (1 + s * np.random.default_rng(5).normal(size=D)), dtype=torch.float64, device=dev,  # noqa: E731
                                           requires_grad=True)
        q0 = torch.tensor(np.where(np.arange(D) % 3 == 0, 0.3, 2.0), dtype=torch.float64, device=dev, requires_grad=True)
        ps = [dict(sigma=v(0.4), rho=v(20.0), Q=q0), dict(sigma=v(0.3), rho=v(10.0), Q=v(1.0)), dict(sigma=v(0.2), rho=v(2.0), Q=v(0.8)),
              dict(sigma=v(0.25), rho=v(5.0), Q=v(1.5)), dict(sigma=v(0.15), rho=v(1.1), Q=v(4.0))]
        kern = xo.gp.terms.SHOTerm(**ps[0])
        for p_ in ps[1:]:
            kern = kern + xo.gp.terms.SHOTerm(**p_)
        leaves = [x for p_ in ps for x in p_.values()]
        with chunks(C):
            ll = xo.gp.GaussianProcess(kern, t=t, yerr=0.2).log_likelihood(y)
            g = torch.autograd.grad(ll.sum(), leaves)
        res[C] = (ll.detach().cpu().numpy(), [x.cpu().numpy() for x in g])
    assert not np.array_equal(res[None][0], res[0][0])
    np.testing.assert_allclose(res[None][0], res[0][0], rtol=1e-12)
    for a, b in zip(res[None][1], res[0][1]):
        assert np.abs(a - b).max() <= 5e-9 * np.abs(b).max() + 1e-300


@pytest.mark.parametrize("name", ["sho_q3", "real1", "three_sho_j6"])
@pytest.mark.parametrize("C", [0, 5, None])
def test_observed_series_minus_model_inside_the_kernels(dev, name, C):
    """exo_celerite_loglike_obs_*: obs - model formed on the fly == the explicit residual array,
    and d/d model == -(d/d resid); sequential, forced-chunk and default plans"""
    from exoplanet_amd.gp import celerite_loglike

    rng = np.random.default_rng(3)
    D, N = 5, 1500
    t = np.sort(rng.uniform(0, 40, N))
    obs = 1e-3 * rng.normal(size=N)
    model = 1e-3 * rng.normal(size=(D, N))
    diag = np.full((1, N), 1e-6)
    cr, cc = batch(rng, name, D)
    with chunks(C):
        want = value_and_grads(dev, t, obs[None] - model, diag, cr, cc)
        tt, mt, dt = T(t, dev), T(model, dev, True), T(diag, dev, True)
        crt, cct = T(cr, dev, True), T(cc, dev, True)
        ll = celerite_loglike(tt, mt, dt, crt, cct, obs=T(obs, dev))
        w = torch.linspace(0.5, 1.5, D, dtype=torch.float64, device=ll.device)
        (ll * w).sum().backward()
    got = [x.detach().cpu().numpy() for x in (ll, mt.grad, dt.grad, crt.grad, cct.grad)]
    np.testing.assert_array_equal(got[0], want[0])           # same arithmetic, same bits
    np.testing.assert_array_equal(got[1], -want[1])
    for a, b in zip(got[2:], want[2:]):
        np.testing.assert_array_equal(a, b)


def test_gaussian_process_mean_model_takes_the_fused_route(dev):
    """GaussianProcess(kernel, t, yerr, mean=model (D, N)).log_likelihood(y (N,)) == the explicit
    y - model, values and gradient with respect to the model"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(4)
    D, N = 3, 2000
    t = T(np.sort(rng.uniform(0, 30, N)), dev)
    y = T(1e-3 * rng.normal(size=N), dev)
    term = xo.gp.terms.SHOTerm(sigma=T(np.full(D, 1e-3), dev), rho=T(np.full(D, 4.0), dev), Q=T(np.full(D, 0.7), dev))
    m1 = T(1e-3 * rng.normal(size=(D, N)), dev, True)
    m2 = m1.detach().clone().requires_grad_(True)
    ll1 = xo.gp.GaussianProcess(term, t=t, yerr=5e-4, mean=m1).log_likelihood(y)
    ll2 = xo.gp.GaussianProcess(term, t=t, yerr=5e-4).log_likelihood(y - m2)
    assert torch.equal(ll1, ll2)
    g1, = torch.autograd.grad(ll1.sum(), m1)
    g2, = torch.autograd.grad(ll2.sum(), m2)
    assert torch.equal(g1, g2)


def test_library_keeps_nothing_between_calls(dev):
    """The forward / reverse calls of a pair share nothing but their arguments (ABI >= 7: the plan is a pure function
    of (n, n_draw, J, n_chunks), the state buffer is the caller's): backward passes run after > 4096 unrelated
    forwards of other shapes and plans, in any order, and under a different EXO_GP_CHUNKS than their forward saw."""
    from exoplanet_amd.gp import celerite_loglike

    rng = np.random.default_rng(77)
    cases = []
    for n, D, name, c in ((700, 3, "sho_q3", 8), (1500, 2, "three_sho_j6", 16), (900, 5, "mixed_j5", None), (400, 2, "real1", 0)):
        t = np.sort(rng.uniform(0, 30, n))
        y, diag = rng.normal(size=(D, n)), 0.1 + 0.1 * rng.uniform(size=(D, n))
        cr, cc = batch(rng, name, D)
        with chunks(0):
            want = value_and_grads(dev, t, y, diag, cr, cc)
        with chunks(c):
            tt, yt, dt = T(t, dev), T(y, dev, True), T(diag, dev, True)
            crt, cct = T(cr, dev, True), T(cc, dev, True)
            ll = celerite_loglike(tt, yt, dt, crt, cct)
        cases.append((ll, (yt, dt, crt, cct), want))
    # > 4096 other forwards (with saved state: requires_grad) in between
    ts = T(np.linspace(0, 5, 96), dev)
    for i in range(4200):
        with chunks((2, 3, None)[i % 3]):
            ys = torch.zeros(1 + i % 2, 96, dtype=torch.float64, device=dev, requires_grad=True)
            celerite_loglike(ts, ys, torch.ones(1, 96, dtype=torch.float64, device=dev),
                             torch.tensor([[[1.0, 0.5]]], dtype=torch.float64, device=dev).expand(1 + i % 2, 1, 2).contiguous(),
                             torch.zeros(1 + i % 2, 0, 4, dtype=torch.float64, device=dev))
    with chunks(5):                      # not what any of the forwards was run under
        for ll, leaves, want in reversed(cases):
            w = torch.linspace(0.5, 1.5, ll.numel(), dtype=torch.float64, device=ll.device)
            (ll * w).sum().backward()
            got = [ll.detach().cpu().numpy()] + [x.grad.cpu().numpy() for x in leaves]
            assert np.allclose(got[0], want[0], rtol=1e-12)
            for a, b in zip(got[1:], want[1:]):
                if b.size:
                    np.testing.assert_allclose(a, b, rtol=2e-7, atol=1e-9 * np.abs(b).max())


@pytest.mark.parametrize("name,D,C,N", [("three_sho_j6", 13, None, 2100), ("mixed_j5", 70, None, 2100), ("sho_q3", 9, 11, 2100),
                                        ("real1", 5, 2, 2100), ("three_sho_j6", 5, 500, 16000), ("mixed_j5", 3, 333, 12000)])
def test_robust_route_whole_batch(dev, name, D, C, N):
    """EVERY draw of the batch above the trees' conditioning thresholds (a signal 3e3 .. 3e4 x the error bars: scores of 1e7 ..
    9e8 / 10 -- under the robust route's 1e8): Newton iterations on the entering states and the adjoint inputs from the chunks' own recurrences
    (docs/DESIGN_r1_r4.md 3.11) for a number of draws that fills no block evenly, state widths 2, 5 and 6, forced chunk counts down to two --
    against the sequential kernels, which since round 4 carry the oscillation-rate gradient as a phase flux too"""
    rng = np.random.default_rng(21)
    t = np.sort(rng.uniform(0, 80 * N / 2100, N))      # (the last two cases: hundreds of chunks -- the depth of the trees whose states the
    if name not in KERNELS:                            # Newton iterations start from)
        pytest.skip("kernel not in the table")
    cr, cc = batch(rng, name, D)
    amp = cr[..., 0].sum(-1) + cc[..., 0].sum(-1)
    ba2 = ((cc[..., 1] / cc[..., 0]) ** 2).max(-1) if cc.shape[1] else np.zeros(D)
    kappa = 10 ** rng.uniform(7.2, 7.9, size=D)          # (J <= 2: above 1e7; wider states: far above 3e4)
    diag = ((1 + ba2) * amp / kappa)[:, None] * (1 + 0.5 * rng.uniform(size=(D, N)))
    y = np.sqrt(amp)[:, None] * rng.normal(size=(D, N))
    with chunks(0):
        want = value_and_grads(dev, t, y, diag, cr, cc)
    with chunks(C):
        got = value_and_grads(dev, t, y, diag, cr, cc)
    assert not np.array_equal(got[0], want[0])                       # (not the sequential kernels' bits: the route was taken)
    # (white noise of the signal's variance against error bars 3e3 .. 9e3 times smaller: log-likelihoods of -1e7 .. -8e7 whose
    # terms carry the conditioning score times the rounding -- two algorithms meet at a few 1e-9 of them)
    np.testing.assert_allclose(got[0], want[0], rtol=3e-9)
    for g, w in zip(got[1:], want[1:]):
        if w.size:
            for d in range(D):
                e = np.abs(g[d] - w[d]).max() / (np.abs(w[d]).max() + 1e-300)
                assert e <= 1e-6, (d, e)


@pytest.mark.parametrize("D,frac", [(130, 0.5), (64, 0.02), (200, 0.97), (7, 0.4)])
def test_mixed_pair_kinds_take_the_draws_kind_by_kind(dev, D, frac):
    """J = 2 with per-draw pair kinds: the chunk kernels' lanes take the draws in the order celerite_kind_partition_kernel
    leaves in the workspace (complex-term draws, padded to whole waves, then two-real-terms draws) -- batches that fill no wave
    evenly, one with a single draw of the other kind, against the sequential kernels draw by draw"""
    from exoplanet_amd.gp import celerite_loglike

    rng = np.random.default_rng(31 + D)
    N = 1500
    t = np.sort(rng.uniform(0, 50, N))
    y = 0.3 * rng.normal(size=(D, N))
    diag = 0.02 + 0.02 * rng.uniform(size=(D, N))
    kind = (rng.uniform(size=D) < frac).astype(np.int32)
    if frac < 0.1:
        kind[:] = 0
        kind[D // 2] = 1
    cplx = np.zeros((D, 1, 4))
    for d in range(D):
        sig, rho = 0.6 * (1 + 0.1 * rng.normal()), 4.0 * (1 + 0.1 * rng.normal())
        if kind[d]:     # two real terms (a1, c1, a2, c2): an over-damped SHO term's
            Q = rng.uniform(0.2, 0.45)
            ar, cr, *_ = P.sho_coefficients(*P.sho_from_sigma_rho(sig, rho, Q), Q)
            cplx[d, 0] = [ar[0], cr[0], ar[1], cr[1]]
        else:
            Q = rng.uniform(0.6, 3.0)
            _, _, ac, bc, cc, dc = P.sho_coefficients(*P.sho_from_sigma_rho(sig, rho, Q), Q)
            cplx[d, 0] = [ac[0], bc[0], cc[0], dc[0]]
    kt = torch.as_tensor(kind[:, None], device=dev)
    res = []
    for c in (1, None):
        yt, dt, ct = T(y, dev, True), T(diag, dev, True), T(cplx, dev, True)
        ll = celerite_loglike(T(t, dev), yt, dt, T(np.zeros((D, 0, 2)), dev), ct, pair_kind=kt, n_chunks=c)
        w = torch.linspace(0.5, 1.5, D, dtype=torch.float64, device=dev)
        (ll * w).sum().backward()
        res.append([x.detach().cpu().numpy() for x in (ll, yt.grad, dt.grad, ct.grad)])
    want, got = res
    assert np.isfinite(want[0]).all()
    np.testing.assert_allclose(got[0], want[0], rtol=1e-11)
    for g, w_ in zip(got[1:], want[1:]):
        for d in range(D):
            assert np.abs(g[d] - w_[d]).max() <= 2e-8 * np.abs(w_[d]).max(), d
