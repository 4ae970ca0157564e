"""GPU: EXO_GP_PREPARE_ADJOINT (include/exoplanet_amd.h) -- the adjoint scan run by the FORWARD call for a cotangent of one, on a
second stream beside the forward chunk kernel, against the serial order of the same pair: the same log-likelihood bits; with a
cotangent of one the same gradient bits; with any other cotangent the same gradients to rounding (the scan is linear in it);
eagerly and inside a captured, replayed graph; with draws on the robust route and draws redone sequentially in the batch."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P
from test_gpu_gp import T

pytestmark = pytest.mark.gpu


def _terms(rng, D, kind):
    """(real [D, Jr, 2], complex [D, Jc, 4], pair_kind or None)"""
    if kind == "sho_mixed":     # J = 2, per-draw pair kinds (Q on both sides of 1/2)
        cplx, pk = np.zeros((D, 1, 4)), np.zeros((D, 1), dtype=np.int32)
        for d in range(D):
            Q = 0.3 if d % 3 == 0 else 0.9
            co = P.sho_coefficients(*P.sho_from_sigma_rho(1e-3 * (1 + 0.1 * d), 4.0 + 0.2 * d, Q), Q)
            if Q < 0.5:
                cplx[d, 0] = [co[0][0], co[1][0], co[0][1], co[1][1]]
                pk[d, 0] = 1
            else:
                cplx[d, 0] = [co[2][0], co[3][0], co[4][0], co[5][0]]
        return np.zeros((D, 0, 2)), cplx, pk
    n_c = {"j6": 3, "j8": 4, "j10": 5}[kind]
    cplx = np.zeros((D, n_c, 4))
    for d in range(D):
        for k in range(n_c):
            co = P.sho_coefficients(*P.sho_from_sigma_rho(1e-3 / (1 + k), 2.0 + 3.0 * k + 0.1 * d, 1.5 + k), 1.5 + k)
            cplx[d, k] = [co[2][0], co[3][0], co[4][0], co[5][0]]
    return np.zeros((D, 0, 2)), cplx, None


def _run(dev, t, y, diag, real, cplx, pk, w, prepare):
    from exoplanet_amd.gp import celerite

    old = celerite._PREPARE[0]
    celerite._PREPARE[0] = prepare
    try:
        yt, ct, dt = T(y, dev, True), T(cplx, dev, True), T(diag, dev, True)
        rt = T(real, dev)
        pkt = None if pk is None else torch.as_tensor(pk, device=dev)
        ll = celerite.celerite_loglike(T(t, dev), yt, dt, rt, ct, pair_kind=pkt)
        (ll * T(w, dev)).sum().backward()
        torch.cuda.synchronize()
        return ll.detach().cpu().numpy(), yt.grad.cpu().numpy(), ct.grad.cpu().numpy(), dt.grad.cpu().numpy()
    finally:
        celerite._PREPARE[0] = old


@pytest.mark.parametrize("kind", ["sho_mixed", "j6", "j8", "j10"])
def test_prepared_adjoint_matches_serial_order(dev, kind):
    rng = np.random.default_rng(5)
    D, n = 70, 6000
    t = np.sort(np.arange(n) * (30.0 / 1440.0) + 1e-3 * rng.uniform(size=n))
    real, cplx, pk = _terms(rng, D, kind)
    y = 5e-4 * rng.normal(size=(D, n))
    diag = np.full((1, n), 2.5e-7)
    for w in (np.ones(D), np.linspace(-0.5, 2.0, D)):
        a = _run(dev, t, y, diag, real, cplx, pk, w, False)
        b = _run(dev, t, y, diag, real, cplx, pk, w, True)
        assert np.array_equal(a[0], b[0])
        if np.all(w == 1.0):
            for x, z in zip(a[1:], b[1:]):
                assert np.array_equal(x, z)
        else:
            for x, z in zip(a[1:], b[1:]):
                scale = np.abs(x).max(axis=tuple(range(1, x.ndim)), keepdims=True) + 1e-300
                assert np.max(np.abs(x - z) / scale) < 1e-12


def test_prepared_adjoint_with_flagged_draws_and_under_graph_replay(dev):
    """a batch with draws on the robust route (high conditioning score) and a draw the filter form does not admit (redone
    sequentially); the pair captured once and replayed: the replays carry the fork"""
    from exoplanet_amd.gp import celerite

    rng = np.random.default_rng(6)
    D, n = 66, 8000
    t = np.arange(n) * (30.0 / 1440.0)
    real, cplx, _ = _terms(rng, D, "j6")
    y = 5e-4 * rng.normal(size=(D, n))
    diag = np.full((D, n), 2.5e-7)
    diag[3] *= 1e-5            # bright draws: high conditioning scores
    diag[40] *= 1e-6
    cplx[17, 1, 1] = 50.0 * cplx[17, 1, 0]     # |b d| > a c: no filter form
    w = np.linspace(0.5, 1.5, D)
    a = _run(dev, t, y, diag, real, cplx, None, w, False)
    b = _run(dev, t, y, diag, real, cplx, None, w, True)
    assert np.array_equal(a[0], b[0], equal_nan=True)
    assert np.isfinite(a[0]).sum() >= D - 1
    for x, z in zip(a[1:], b[1:]):
        assert np.array_equal(np.isfinite(x), np.isfinite(z))
        x, z = np.nan_to_num(x, posinf=0.0, neginf=0.0), np.nan_to_num(z, posinf=0.0, neginf=0.0)
        scale = np.abs(x).max(axis=tuple(range(1, x.ndim)), keepdims=True) + 1e-300
        assert np.max(np.abs(x - z) / scale) < 1e-10
    # captured and replayed
    old = celerite._PREPARE[0]
    celerite._PREPARE[0] = True
    try:
        tt, yt, ct, dt, rt, wt = T(t, dev), T(y, dev, True), T(cplx, dev, True), T(diag, dev, True), T(real, dev), T(w, dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                ll = celerite.celerite_loglike(tt, yt, dt, rt, ct)
                gy, gc = torch.autograd.grad((ll * wt).sum(), (yt, ct))
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ll = celerite.celerite_loglike(tt, yt, dt, rt, ct)
            gy, gc = torch.autograd.grad((ll * wt).sum(), (yt, ct))
        for _ in range(3):
            ll.detach().zero_(); gy.zero_(); gc.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(ll.detach().cpu().numpy(), b[0], equal_nan=True)
            assert np.array_equal(gy.cpu().numpy(), b[1], equal_nan=True)
            assert np.array_equal(gc.cpu().numpy(), b[2], equal_nan=True)
    finally:
        celerite._PREPARE[0] = old
