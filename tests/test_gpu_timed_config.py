"""GPU parity of the EXACT configurations bench.py times, at the sizes it times them.

bench.py's own step builders (`bench.workload_c2` .. `workload_c5`, `sparse_step_fn`, `likelihood_step_fn`) are imported,
captured as a hipGraph exactly as the timed region does (`exoplanet_amd.GraphedStep`) and replayed; the replay's outputs
are held to the oracle's C port (oracle/c, OpenMP over draws) DIRECTLY:

* C2, D = 1024 draws x 150 000 cadences (one block per draw: the heavy kernel finishes its own draws; `flux_dot` with the
  column-form packing VJP): the dense flux of EVERY draw, the per-draw scalar, all 8 leaf gradients of every draw;
* the same batch through the sparse step and the white-noise-likelihood step (what the sampler legs run);
* C4 at 64 draws, C5 at 128 chains (light curve of every chain; log-likelihood and gradients of >= 8 chains) -- the per-GPU
  shares of an 8-GPU run -- AND at 512 draws / 1024 chains, the shapes `bench.py --config c4|c5 --gpus 1` times;
  C3 at 1024 draws (log-likelihood of EVERY draw, gradients of >= 8 draws);
* C1's exact size (10 000 cadences, circular) and a `duration=` orbit, end to end on the HIP path.

Leaf gradients of the oracle: its record / limb-darkening cotangents (C port) chained through the Jacobian of the numpy
restatement of the reference's constructor algebra (oracle/numpy_port.KeplerianOrbit, keplerian.py:75-281; five-point
central differences, relative step 1e-4: truncation ~1e-16, rounding ~1e-12).

reference: src/exoplanet/light_curves/limb_dark.py:99-232, src/exoplanet/orbits/keplerian.py:849-934,
src/exoplanet/light_curves/secondary_eclipse.py:33-70.
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import c_port as C
from oracle import numpy_port as P

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ORBIT_KEYS = ("period", "t0", "b", "ecc", "omega")
GRAD_REL = 2e-7      # leaf gradients: relative to the largest gradient of that leaf over the batch
# ... and PER DRAW (round 6) for the light-curve configs: 5e-6 of the draw's own gradient (floor: 1e-3 of the
# batch's largest).  Not 1e-6: the limit is the ORACLE's -- its analytic VJP chained to the leaves is itself right to ~2e-7 of a
# draw's gradient (tools/grad_diag.py: Richardson differences of the oracle's own L agree with the HIP gradient to 1e-9 where the
# oracle's chain is 2e-7 off), and a draw whose random-sign sum passes near zero multiplies that; measured worst 2.2e-6.
GRAD_REL_PER_DRAW_LC = 5e-6
FLUX_ABS = 1e-12     # north-star gate is 1e-6 relative flux error


def npy(x):
    return x.detach().cpu().numpy()


def oracle_threads():
    import bench

    n = bench.usable_cpus()[0]
    C.lib().oracle_set_threads(int(n))
    return n


def records(vals, sbr=None):
    """numpy-port orbit -> kernel records (D, Pn, NPAR) for leaf values given as (D, Pn) arrays (every (draw, planet) is an
    independent orbit: m_planet = 0), as tests/test_gpu_transit.make_record does for one draw"""
    D, Pn = vals["period"].shape
    o = P.KeplerianOrbit(**{k: vals[k].ravel() for k in ORBIT_KEYS if k in vals})
    n = D * Pn
    rec = np.zeros((n, P.NPAR))
    circ = o.ecc is None
    rec[:, P.P_N], rec[:, P.P_TP] = o.n, o.t_periastron
    rec[:, P.P_ECC] = 0.0 if circ else o.ecc
    rec[:, P.P_COSW] = 1.0 if circ else o.cos_omega
    rec[:, P.P_SINW] = 0.0 if circ else o.sin_omega
    rec[:, P.P_COSI], rec[:, P.P_SINI] = o.cos_incl, o.sin_incl
    rec[:, P.P_AOR], rec[:, P.P_ROR] = o.a / o.r_star, vals["r"].ravel() / o.r_star
    rec[:, P.P_T0], rec[:, P.P_PERIOD] = o.t0, o.period
    rec[:, P.P_TS], rec[:, P.P_TE], rec[:, P.P_TS2], rec[:, P.P_TE2] = -np.inf, np.inf, -np.inf, np.inf
    if sbr is not None:
        rec[:, P.P_FRATIO] = (sbr[:, None] * np.ones((D, Pn))).ravel() * rec[:, P.P_ROR] ** 2
    return rec.reshape(D, Pn, P.NPAR)


def chain_to_leaves(vals, gparams, slots, sbr=None, gld=None, u=None):
    """oracle cotangents of the records (and of the Green's-basis coefficients) -> cotangents of the leaves"""
    D, Pn = vals["period"].shape
    out = {}
    wrt = [k for k in vals] + (["sbr"] if sbr is not None else [])
    for k in wrt:
        x0 = sbr if k == "sbr" else vals[k]
        h = 1e-4 * np.maximum(np.abs(x0), 1e-2)

        def at(s):
            if k == "sbr":
                return records(vals, sbr + s * h)
            v = dict(vals)
            v[k] = x0 + s * h
            return records(v, sbr)

        hh = h[:, None, None] if k == "sbr" else h[..., None]
        with np.errstate(invalid="ignore"):      # (the window slots hold +-inf: never among `slots`)
            J = (-at(2.0) + 8.0 * at(1.0) - 8.0 * at(-1.0) + at(-2.0)) / (12.0 * hh)
        g = (gparams[..., slots] * J[..., slots]).sum(-1)
        out[k] = g.sum(-1) if k == "sbr" else g
    if gld is not None:
        u1, u2 = u
        for k, (d1, d2) in (("u1", (1.0, 0.0)), ("u2", (0.0, 1.0))):
            h = 1e-4
            c = lambda s: np.stack(P.get_cl(u1 + s * h * d1, u2 + s * h * d2), -1)  # noqa: E731
            J = (-c(2.0) + 8.0 * c(1.0) - 8.0 * c(-1.0) + c(-2.0)) / (12.0 * h)
            out[k] = (gld[:, :3] * J).sum(-1)
    return out


def assert_grads(got, want, names, rel=GRAD_REL, rows=None, per_draw=False):
    """per_draw (the GP configs): every checked draw's gradient is held to `rel` of ITS OWN magnitude -- BASELINE.md's
    1e-6 -- not to `rel` of the largest gradient in the batch; a draw whose gradient passes through zero is measured against
    1e-3 of the batch's largest instead (an absolute floor of rel x 1e-3 x scale)"""
    for k in names:
        g, w = npy(got[k]).reshape(want[k].shape if rows is None else (-1,) + want[k].shape[1:]), want[k]
        if rows is not None:
            g = g[rows]
        scale = np.abs(w).max()
        assert scale > 0, k
        if per_draw:
            den = np.maximum(np.abs(w), 1e-3 * scale)
            worst = (np.abs(g - w) / den).max()
            assert worst <= rel, (k, worst)
        else:
            err = np.abs(g - w).max()
            assert err <= rel * scale, (k, err / scale)


# ---------------------------------------------------------------------------------------------------
# C2 at the timed size: D = 1024, N = 150 000, hipGraph replay
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2(dev):
    """bench.py's C2 workload at its default size + the oracle's flux / cotangents for every draw (computed once)"""
    import bench
    import exoplanet_amd as xo
    from exoplanet_amd import ops

    D = 1024
    wl = bench.workload_c2(xo, ops, dev, D, rank=0)
    oracle_threads()
    vals = {k: npy(v).reshape(D, 1) for k, v in zip(wl.names, wl.leaves) if k not in ("u1", "u2")}
    u1, u2 = (npy(dict(zip(wl.names, wl.leaves))[k]) for k in ("u1", "u2"))
    rec = records(vals)
    c = np.stack(P.get_cl(u1, u2), -1)
    t, g = npy(wl.data["t"]), npy(wl.data["gbar"])
    want_f, want_gp, want_gl = C.transit(t, rec, c, g)
    sl = list(P.GRAD_SLOTS[:-1])
    leaf = chain_to_leaves(vals, want_gp, sl, gld=want_gl, u=(u1, u2))
    return dict(wl=wl, xo=xo, ops=ops, bench=bench, D=D, vals=vals, u=(u1, u2), rec=rec, c=c, t=t, g=g, f=want_f, gp=want_gp,
                gl=want_gl, leaf=leaf, sl=sl)


def test_c2_timed_step_every_draw_vs_oracle(dev, c2):
    """the step `bench.py` times (default flags), replayed as a hipGraph: flux of all 1024 draws x 150 000 cadences, the
    per-draw scalar and the 8 leaf gradients of every draw against oracle/c"""
    wl, xo = c2["wl"], c2["xo"]
    graph = xo.GraphedStep(wl.fn, *wl.leaves)
    for _ in range(3):                      # the timed loop replays it over and over: so does the test
        out = graph()
    torch.cuda.synchronize()
    flux, L, grads = out[0], out[1], dict(zip(wl.names, out[2:]))
    assert flux.shape == (c2["D"], 150_000)
    f = npy(flux)
    assert (c2["f"] < -1e-3).sum() > 3000 * c2["D"]            # the transits are there
    assert np.abs(f - c2["f"]).max() < FLUX_ABS
    want_L = (c2["g"] * c2["f"]).sum(-1)
    assert np.abs(npy(L) - want_L).max() <= 1e-10 * np.abs(want_L).max()
    assert_grads(grads, {k: (v if v.ndim == 1 else v) for k, v in c2["leaf"].items()}, wl.names)
    # ... and every draw's gradient relative to ITS OWN magnitude, as the GP configs are held
    assert_grads(grads, c2["leaf"], wl.names, rel=GRAD_REL_PER_DRAW_LC, per_draw=True)
    # the eager step (what the hipEvent-timed launches of bench.py run) is the same computation
    eager = wl.fn(*wl.leaves)
    assert torch.equal(eager[0], flux) and torch.equal(eager[1], L)
    for a, b in zip(eager[2:], out[2:]):
        assert torch.equal(a, b)


def test_c2_sparse_step_vs_oracle(dev, c2):
    """`extras.c2_sparse_output` at D = 1024: per-draw scalar + leaf gradients of every draw; the runs + values rebuilt to
    the dense light curve for a few draws"""
    wl, xo, ops, bench = c2["wl"], c2["xo"], c2["ops"], c2["bench"]
    fn = bench.sparse_step_fn(xo, ops, wl.names, wl.data["t"], wl.data["gbar"])
    graph = xo.GraphedStep(fn, *wl.leaves)
    out = graph()
    torch.cuda.synchronize()
    want_L = (c2["g"] * c2["f"]).sum(-1)
    assert np.abs(npy(out[0]) - want_L).max() <= 1e-10 * np.abs(want_L).max()
    assert_grads(dict(zip(wl.names, out[1:])), c2["leaf"], wl.names)
    assert_grads(dict(zip(wl.names, out[1:])), c2["leaf"], wl.names, rel=GRAD_REL_PER_DRAW_LC, per_draw=True)
    lv = dict(zip(wl.names, wl.leaves))
    orbit = xo.KeplerianOrbit(**{k: lv[k].detach() for k in ORBIT_KEYS})
    rec, ld, _, flags = orbit.kernel_inputs(lv["r"].detach(), (lv["u1"].detach(), lv["u2"].detach()), use_in_transit=False)
    rows = [0, 511, 1023]
    sp = ops.transit_flux_sparse(wl.data["t"], rec[rows].contiguous(), ld[rows].contiguous(), flags=flags)
    assert np.abs(sp.to_dense() - c2["f"][rows]).max() < FLUX_ABS


def test_c2_white_noise_likelihood_step_vs_oracle(dev, c2):
    """`extras.c2_white_noise_likelihood` at D = 1024 (one evaluation per solved cadence, no (draw, cadence) array):
    log-likelihood and leaf gradients of every draw.  The oracle's gradient: the C port's VJP with the cotangent
    w (y - f) of its own flux"""
    wl, xo, bench = c2["wl"], c2["xo"], c2["bench"]
    N, yerr = 150_000, 1e-4
    y = 1e-4 * np.random.default_rng(12).normal(size=N) + c2["f"][0]       # data = the first draw's curve + noise
    obs = torch.as_tensor(y, device=dev)
    graph = xo.GraphedStep(bench.likelihood_step_fn(xo, wl.names, wl.data["t"], obs, yerr), *wl.leaves)
    out = graph()
    torch.cuda.synchronize()
    w = 1.0 / yerr ** 2
    r = y[None, :] - c2["f"]
    want_ll = -0.5 * w * (r * r).sum(-1) + 0.5 * N * np.log(w / (2 * np.pi))
    assert np.abs(npy(out[0]) - want_ll).max() <= 1e-10 * np.abs(want_ll).max()
    _, gp, gl = C.transit(c2["t"], c2["rec"], c2["c"], w * r, want_flux=False)
    want = chain_to_leaves(c2["vals"], gp, c2["sl"], gld=gl, u=c2["u"])
    assert_grads(dict(zip(wl.names, out[1:])), want, wl.names, rel=1e-6)     # (sums of ~4000 terms of alternating sign)


def test_c2_per_draw_jitter_gradient(dev, c2):
    """a per-draw error bar (a jitter term sampled per chain) gets its FULL gradient through
    white_noise_log_likelihood; a per-cadence error bar that requires grad takes the dense path (never a partial one)"""
    wl, xo = c2["wl"], c2["xo"]
    D, N = 8, 150_000
    lv = {k: v.detach()[:D].clone().requires_grad_(True) for k, v in zip(wl.names, wl.leaves)}
    y = torch.as_tensor(1e-4 * np.random.default_rng(13).normal(size=N) + c2["f"][0], device=dev)
    jit = torch.full((D, 1), 1.3e-4, dtype=torch.float64, device=dev, requires_grad=True)
    orbit = xo.KeplerianOrbit(**{k: lv[k] for k in ORBIT_KEYS})
    ll = xo.LimbDarkLightCurve(lv["u1"], lv["u2"]).white_noise_log_likelihood(orbit=orbit, r=lv["r"], t=wl.data["t"], y=y, yerr=jit)
    (gj,) = torch.autograd.grad(ll.sum(), jit)
    r = npy(y)[None, :] - c2["f"][:D]
    s = 1.3e-4
    want_ll = -0.5 * (r * r).sum(-1) / s ** 2 - N * np.log(s) - 0.5 * N * np.log(2 * np.pi)
    want_g = (r * r).sum(-1) / s ** 3 - N / s
    assert np.abs(npy(ll) - want_ll).max() <= 1e-10 * np.abs(want_ll).max()
    assert np.abs(npy(gj)[:, 0] - want_g).max() <= 1e-9 * np.abs(want_g).max()
    # per-cadence error bars that require grad: dense fallback, full gradient
    ye = torch.full((N,), 1.3e-4, dtype=torch.float64, device=dev, requires_grad=True)
    orbit = xo.KeplerianOrbit(**{k: lv[k] for k in ORBIT_KEYS})
    ll2 = xo.LimbDarkLightCurve(lv["u1"], lv["u2"]).white_noise_log_likelihood(orbit=orbit, r=lv["r"], t=wl.data["t"], y=y, yerr=ye)
    (ge,) = torch.autograd.grad(ll2.sum(), ye)
    assert np.abs(npy(ll2) - want_ll).max() <= 1e-10 * np.abs(want_ll).max()
    want_ge = ((r * r) / s ** 3 - 1.0 / s).sum(0)
    assert np.abs(npy(ge) - want_ge).max() <= 1e-9 * np.abs(want_ge).max()
    # the op itself refuses what it cannot differentiate (nothing is dropped silently)
    from exoplanet_amd import ops
    rec, ld, _, flags = orbit.kernel_inputs(lv["r"], (lv["u1"], lv["u2"]), use_in_transit=False)
    with pytest.raises(NotImplementedError):
        ops.white_noise_loglike(wl.data["t"], rec, ld, y, ye, flags=flags)
    with pytest.raises(NotImplementedError):
        ops.transit_chi2(wl.data["t"], rec, ld, y, (1.0 / (ye * ye)), flags=flags)


# ---------------------------------------------------------------------------------------------------
# C4 at 64 draws
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [64, 512])
def test_c4_timed_step_vs_oracle(dev, D):
    """`bench.py --config c4` on one of eight GPUs / `extras.c4_four_planets_64_draws` (64 draws) and on ONE GPU
    (`--config c4 --gpus 1`: all 512 draws of BASELINE's total -- the single-GPU point of the strong-scaling curve, where a
    block finishes its own draw): 4 planets, 200 000 cadences, hipGraph replay; flux of every draw, per-draw scalar, all
    4 x 6 + 2 leaf gradients per draw"""
    import bench
    import exoplanet_amd as xo
    from exoplanet_amd import ops

    wl = bench.workload_c4(xo, ops, dev, D, rank=0)
    oracle_threads()
    lv = dict(zip(wl.names, wl.leaves))
    vals = {k: npy(lv[k]) for k in wl.names if k not in ("u1", "u2")}
    u1, u2 = npy(lv["u1"]), npy(lv["u2"])
    rec, c = records(vals), np.stack(P.get_cl(u1, u2), -1)
    t, g = npy(wl.data["t"]), npy(wl.data["gbar"])
    want_f, want_gp, want_gl = C.transit(t, rec, c, g)
    graph = xo.GraphedStep(wl.fn, *wl.leaves)
    out = graph()
    torch.cuda.synchronize()
    assert (want_f < -1e-3).sum() > 5000 * D
    assert np.abs(npy(out[0]) - want_f).max() < FLUX_ABS
    want_L = (g * want_f).sum(-1)
    assert np.abs(npy(out[1]) - want_L).max() <= 1e-10 * np.abs(want_L).max()
    want = chain_to_leaves(vals, want_gp, list(P.GRAD_SLOTS[:-1]), gld=want_gl, u=(u1, u2))
    assert_grads(dict(zip(wl.names, out[2:])), want, wl.names)
    assert_grads(dict(zip(wl.names, out[2:])), want, wl.names, rel=GRAD_REL_PER_DRAW_LC, per_draw=True)      # per (draw, planet)


# ---------------------------------------------------------------------------------------------------
# GP configs: C3 at 1024 draws, C5 at 128 chains
# ---------------------------------------------------------------------------------------------------
def sho_coeffs(terms):
    parts = [P.sho_coefficients(*P.sho_from_sigma_rho(s, rho, q), q) for s, rho, q in terms]
    return tuple(np.concatenate(x) for x in zip(*parts))


def gp_oracle_draw(t, y, diag, flux, terms, wrt):
    """C port: log-likelihood of y - flux under the SHO terms, its cotangent w.r.t. the flux, and its gradient w.r.t. the
    hyper-parameters named in `wrt` = [(term index, 0 sigma | 1 rho | 2 Q)] (coefficient cotangents chained through
    five-point differences of the SHO coefficient algebra)"""
    co = sho_coeffs(terms)
    ll, gw = C.celerite(t, y - flux, diag, co, grad=True)
    gcoef = np.concatenate([gw[k] for k in ("ar", "cr", "ac", "bc", "cc", "dc")])
    ghyper = []
    for ti, which in wrt:
        x0 = terms[ti][which]
        h = 1e-4 * abs(x0)

        def at(s):
            tt = [list(x) for x in terms]
            tt[ti][which] = x0 + s * h
            return np.concatenate(sho_coeffs(tt))

        J = (-at(2.0) + 8.0 * at(1.0) - 8.0 * at(-1.0) + at(-2.0)) / (12.0 * h)
        ghyper.append(float((gcoef * J).sum()))
    return ll, -gw["y"], ghyper


def test_c3_timed_step_vs_oracle(dev):
    """`extras.c3_light_curve_plus_sho_gp` / `bench.py --config c3` at D = 1024: the replayed step's log-likelihood and
    every leaf gradient (8 orbit / limb-darkening leaves + sigma, rho, Q) for 8 draws spread over the batch against the C
    port (light curve -> celerite -> back), and the log-likelihood of ALL 1024 draws against it"""
    import bench
    import exoplanet_amd as xo
    from exoplanet_amd import ops

    D, N = 1024, 150_000
    wl = bench.workload_c3(xo, ops, dev, D, rank=0)
    oracle_threads()
    graph = xo.GraphedStep(wl.fn, *wl.leaves)
    out = graph()
    torch.cuda.synchronize()
    ll, grads = npy(out[0]), dict(zip(wl.names, out[1:]))
    assert np.isfinite(ll).all() and np.unique(ll).size > D // 2
    lv = dict(zip(wl.names, wl.leaves))
    t, y = npy(wl.data["t"]), npy(wl.data["yobs"])
    diag = np.full(N, wl.data["yerr"] ** 2)
    # EVERY draw's log-likelihood against the C port (8 of 1024 were compared): light curve of all 1024 draws
    # (OpenMP over draws), then one sequential celerite pass per draw on a thread pool (ctypes releases the GIL)
    all_vals = {k: npy(lv[k]).reshape(D, 1) for k in ORBIT_KEYS + ("r",)}
    all_flux, _, _ = C.transit(t, records(all_vals), np.stack(P.get_cl(npy(lv["u1"]), npy(lv["u2"])), -1), None)
    hyper = [(float(a), float(b), float(q)) for a, b, q in zip(npy(lv["sigma"]), npy(lv["rho"]), npy(lv["Q"]))]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, oracle_threads())) as pool:
        all_ll = np.array(list(pool.map(lambda d: C.celerite(t, y - all_flux[d], diag, sho_coeffs([hyper[d]])), range(D))))
    assert np.abs(ll - all_ll).max() <= 1e-10 * np.abs(all_ll).max(), np.abs(ll - all_ll).max() / np.abs(all_ll).max()
    rows = np.array([0, 1, 2, 3, 500, 777, 1022, 1023])
    vals = {k: v[rows] for k, v in all_vals.items()}
    u1, u2 = npy(lv["u1"])[rows], npy(lv["u2"])[rows]
    rec, c = records(vals), np.stack(P.get_cl(u1, u2), -1)
    flux = all_flux[rows]
    del all_flux
    want_ll, gflux, gh = [], [], []
    for i, d in enumerate(rows):
        terms = [(float(npy(lv["sigma"])[d]), float(npy(lv["rho"])[d]), float(npy(lv["Q"])[d]))]
        a, b, h = gp_oracle_draw(t, y, diag, flux[i], terms, [(0, 0), (0, 1), (0, 2)])
        want_ll.append(a); gflux.append(b); gh.append(h)
    want_ll, gh = np.array(want_ll), np.array(gh)
    assert np.abs(ll[rows] - want_ll).max() <= 1e-10 * np.abs(want_ll).max()
    _, gp, gl = C.transit(t, rec, c, np.stack(gflux), want_flux=False)
    want = chain_to_leaves(vals, gp, list(P.GRAD_SLOTS[:-1]), gld=gl, u=(u1, u2))
    for k, j in (("sigma", 0), ("rho", 1), ("Q", 2)):
        want[k] = gh[:, j]
    assert_grads(grads, want, wl.names, rel=1e-6, rows=rows, per_draw=True)


@pytest.mark.parametrize("D,bright", [(128, 0), (1024, 0), (128, 2)])
def test_c5_timed_step_vs_oracle(dev, D, bright):
    """`extras.c5_secondary_eclipse_3term_gp_128_chains` / `bench.py --config c5` on one of eight GPUs (128 chains) and on
    ONE GPU (`--config c5 --gpus 1`: all 1024 chains -- another chunk plan, and the light-curve blocks finish their own
    draws): 65 000 long cadences x 7 sub-exposures, transit + occultation, three SHO terms (J = 6).  The light curve of
    EVERY chain against the C port; the replayed step's log-likelihood and all 10 leaf gradients for 8 chains.
    bright = 2: `extras.c5_128_chains_1pct_bright_star_kappa_1e6` -- chains 0 and 1 at a conditioning score of 1e6, finished by
    the ROBUST route of the time-parallel path (docs/DESIGN_r1_r4.md 3.11) at the benchmark's own size: 512 chunks, the serial forward
    chain, the adjoint inputs from the chunks' own recurrences -- both among the chains checked"""
    import bench
    import exoplanet_amd as xo
    from exoplanet_amd import ops

    N = bench.C5_NCAD
    wl = bench.workload_c5(xo, ops, dev, D, rank=0, bright=bright)
    oracle_threads()
    lv = dict(zip(wl.names, wl.leaves))
    vals = {k: npy(lv[k]).reshape(D, 1) for k in ORBIT_KEYS + ("r",)}
    sbr = npy(lv["sbr"])
    rec = records(vals, sbr)
    c = np.repeat(np.concatenate([P.get_cl(*bench.C5_LD[0]), P.get_cl(*bench.C5_LD[1])])[None], D, 0)
    t, y = npy(wl.data["t"]), npy(wl.data["yobs"])
    sdt, sw = P.exposure_stencil(7, 0)
    kw = dict(texp=bench.C5_TEXP, stencil_dt=sdt, stencil_w=sw, secondary=True)
    want_f, _, _ = C.transit(t, rec, c, None, **kw)
    with torch.no_grad():
        orbit = xo.KeplerianOrbit(**{k: lv[k].detach() for k in ORBIT_KEYS})
        lc = xo.SecondaryEclipseLightCurve(bench.C5_LD[0], bench.C5_LD[1], lv["sbr"].detach()).get_light_curve(
            orbit=orbit, r=lv["r"].detach(), t=wl.data["t"], texp=bench.C5_TEXP, oversample=7, total=True)
    assert (want_f < -1e-3).sum() > 1000 * D
    assert np.abs(npy(lc) - want_f).max() < FLUX_ABS
    graph = xo.GraphedStep(wl.fn, *wl.leaves)
    out = graph()
    torch.cuda.synchronize()
    ll, grads = npy(out[0]), dict(zip(wl.names, out[1:]))
    assert np.isfinite(ll).all()
    rows = np.array([0, 1, 2, 63, 64, 100, 126, 127]) if D == 128 else np.array([0, 1, 127, 128, 511, 512, 1000, 1023])
    diag = np.full(N, wl.data["yerr"] ** 2)
    want_ll, gflux, gh = [], [], []
    for d in rows:
        terms = [(float(npy(lv[f"s{j + 1}"])[d]), bench.C5_TERMS[j][1], bench.C5_TERMS[j][2]) for j in range(3)]
        a, b, h = gp_oracle_draw(t, y, diag, want_f[d], terms, [(0, 0), (1, 0), (2, 0)])
        want_ll.append(a); gflux.append(b); gh.append(h)
    want_ll, gh = np.array(want_ll), np.array(gh)
    assert np.abs(ll[rows] - want_ll).max() <= 1e-10 * np.abs(want_ll).max()
    # ... and the log-likelihood of EVERY chain (8 of 128 / 1024 were compared): one sequential celerite pass
    # per chain of the C port on a thread pool (ctypes releases the GIL)
    from concurrent.futures import ThreadPoolExecutor

    def chain_ll(d):
        terms = [(float(npy(lv[f"s{j + 1}"])[d]), bench.C5_TERMS[j][1], bench.C5_TERMS[j][2]) for j in range(3)]
        return C.celerite(t, y - want_f[d], diag, sho_coeffs(terms))
    with ThreadPoolExecutor(max(1, oracle_threads())) as pool:
        all_ll = np.array(list(pool.map(chain_ll, range(D))))
    assert np.abs(ll - all_ll).max() <= 1e-10 * np.abs(all_ll).max(), np.abs(ll - all_ll).max() / np.abs(all_ll).max()
    sub = {k: v[rows] for k, v in vals.items()}
    _, gp, gl = C.transit(t, rec[rows], c[rows], np.stack(gflux), want_flux=False, **kw)
    want = chain_to_leaves(sub, gp, list(P.GRAD_SLOTS), sbr=sbr[rows])
    for j in range(3):
        want[f"s{j + 1}"] = gh[:, j]
    if bright:
        # the chains at a conditioning score of 1e6 (robust route): every gradient within 2e-6 of the largest over the checked
        # chains -- their own gradient with respect to the faint terms' amplitudes is ~1e-2 of that, and two double-precision
        # algorithms differ by kappa x 1e-16 of the LARGE terms there -- and the clean chains to 1e-6 each, as below
        assert_grads(grads, want, wl.names, rel=2e-6, rows=rows)
        clean = rows >= bright
        assert_grads(grads, {k: v[clean] for k, v in want.items()}, wl.names, rel=1e-6, rows=rows[clean], per_draw=True)
    else:
        assert_grads(grads, want, wl.names, rel=1e-6, rows=rows, per_draw=True)


# ---------------------------------------------------------------------------------------------------
# C1's exact size and the duration parameterisation, end to end on the HIP path
# ---------------------------------------------------------------------------------------------------
def test_c1_exact_size_on_the_hip_path(dev):
    """BASELINE configs[0] (C1): circular orbit, 10 000 two-minute cadences, P = 3.5, t0 = 1, b = 0.3, r = 0.1,
    (u1, u2) = (0.3, 0.2) -- through xo.KeplerianOrbit + LimbDarkLightCurve.get_light_curve on the GPU: flux against the
    numpy restatement of the reference glue and against the C port, gradients of the five leaves against the oracle"""
    import exoplanet_amd as xo

    t = np.arange(10_000) * (2.0 / 1440.0)
    base = dict(period=3.5, t0=1.0, b=0.3, r=0.1, u1=0.3, u2=0.2)
    lv = {k: torch.tensor([v], dtype=torch.float64, device=dev, requires_grad=True) for k, v in base.items()}
    orbit = xo.KeplerianOrbit(period=lv["period"], t0=lv["t0"], b=lv["b"])
    lc = xo.LimbDarkLightCurve(lv["u1"][0], lv["u2"][0]).get_light_curve(orbit=orbit, r=lv["r"], t=torch.as_tensor(t, device=dev))
    assert lc.shape == (10_000, 1)
    want = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3), r=0.1, t=t,
                                                         use_in_transit=False)
    assert (want < -1e-3).sum() > 200
    assert np.abs(npy(lc) - want).max() < 1e-13
    for uit in (True, False):       # the reference's default selects in-transit cadences: same curve
        lc2 = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=xo.KeplerianOrbit(period=3.5, t0=1.0, b=0.3), r=0.1,
                                                               t=torch.as_tensor(t, device=dev), use_in_transit=uit)
        assert np.abs(npy(lc2) - want).max() < 1e-13
    vals = {k: np.array([[base[k]]]) for k in ("period", "t0", "b", "r")}
    rec, c = records(vals), P.get_cl(0.3, 0.2)[None]
    g = np.random.default_rng(1).normal(size=(1, t.size))
    f, gp, gl = C.transit(t, rec, c, g)
    assert np.abs(npy(lc)[:, 0] - f[0]).max() < 1e-13
    grads = torch.autograd.grad((lc[:, 0] * torch.as_tensor(g[0], device=dev)).sum(), list(lv.values()))
    wantg = chain_to_leaves(vals, gp, list(P.GRAD_SLOTS[:-1]), gld=gl, u=(np.array([0.3]), np.array([0.2])))
    for k, gg in zip(lv, grads):
        w = float(np.ravel(wantg[k])[0])
        assert abs(float(gg) - w) <= 2e-7 * max(abs(w), 1e-3), (k, float(gg), w)


def test_duration_orbit_on_the_hip_path(dev):
    """KeplerianOrbit(duration=...) (keplerian.py:112-131,237-260; reference tests keplerian_test.py:611-643 with the same
    numbers) end to end on the GPU: the light curve against the oracle's restatement of the same constructor branch, the
    separation at +- duration / 2 = 1 + ror, and d(flux)/d(duration) against central differences of the oracle"""
    import exoplanet_amd as xo

    duration, period, b, ror, r_star = 0.12, 10.1235, 0.34, 0.06, 0.7
    t = np.linspace(-0.2, 0.2, 4001)
    td = torch.as_tensor(t, device=dev)
    dv = torch.tensor(duration, dtype=torch.float64, device=dev, requires_grad=True)
    orbit = xo.KeplerianOrbit(period=period, t0=0.0, b=b, duration=dv, r_star=r_star, ror=ror)
    lc = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=ror * r_star, t=td)

    def oracle(d):
        o = P.KeplerianOrbit(period=period, t0=0.0, b=b, duration=d, r_star=r_star, ror=ror)
        return P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=o, r=ror * r_star, t=t, use_in_transit=False)

    want = oracle(duration)
    assert np.abs(npy(lc) - want).max() < 1e-13
    inside = np.abs(t) < 0.5 * duration - 1e-3
    assert (want[inside] < 0).all() and (want[np.abs(t) > 0.5 * duration + 1e-3] == 0).all()
    x, y, z = orbit.get_planet_position(torch.tensor([-0.5 * duration, 0.5 * duration, period + 0.5 * duration], device=dev))
    assert np.allclose(npy(torch.sqrt(x ** 2 + y ** 2)).ravel(), r_star * (1 + ror))      # (the reference's own tolerance)
    # d flux / d duration is singular (~ 1 / sqrt) at the contact points: central differences of the oracle are held
    # tightly on the cadences away from them, loosely on all
    w = np.random.default_rng(3).normal(size=want.shape)
    away = (np.abs(np.abs(t) - 0.5 * duration) > 6e-3)[:, None]
    for ww, tol in ((w * away, 1e-5), (w, 1e-2)):
        (g,) = torch.autograd.grad((lc * torch.as_tensor(ww, device=dev)).sum(), dv, retain_graph=True)
        h = 1e-6
        fd = ((oracle(duration + h) - oracle(duration - h)) * ww).sum() / (2 * h)
        assert abs(float(g) - fd) <= tol * abs(fd), (float(g), fd)
    # the eccentric branch: duration -> b (keplerian.py:237-260), then the same kernels
    kw = dict(period=5.0, t0=0.2, ecc=0.3, omega=0.7, duration=0.09)
    t2 = np.linspace(-0.1, 0.5, 3001)
    got = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=xo.KeplerianOrbit(**kw), r=0.08, t=torch.as_tensor(t2, device=dev))
    want2 = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=P.KeplerianOrbit(**kw), r=0.08, t=t2, use_in_transit=False)
    assert want2.min() < -1e-3
    assert np.abs(npy(got) - want2).max() < 1e-13
