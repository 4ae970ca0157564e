"""Randomised stress of the timing-variation path of the fused light-curve kernels against the
oracle's numpy evaluation: random batch shapes, planet counts, transit counts (1..40), timing
offsets up to a fifth of a period, missing transits, arbitrary (non-physical) tables, exposures
long enough to cross bin edges, unsorted times, caller windows, secondary eclipses.
    python tools/ttv_stress.py <seed> <cases>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from exoplanet_amd import ops
from oracle import numpy_port as P
from test_gpu_transit import make_record

dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = dict(flux=0.0, gp=0.0, gl=0.0, gs=0.0)
for case in range(n_cases):
    D, Pn = int(rng.integers(1, 7)), int(rng.integers(1, 4))
    N = int(rng.integers(50, 3000))
    secondary, per_planet = rng.uniform() < 0.25, rng.uniform() < 0.4
    window = rng.uniform() < 0.3
    use_texp = rng.uniform() < 0.5
    # tables that no orbit would produce -- not together with caller windows: a window is a statement
    # about the warped mid-exposure time, and with unrelated shifts on either side of an edge a
    # sub-exposure can be in transit while its cadence is outside every window (the reference would
    # then depend on which OTHER planets happen to be in transit, keplerian.py:771)
    arbitrary = rng.uniform() < 0.25 and not window
    span = 10 ** rng.uniform(1.0, 2.3)
    t = np.sort(rng.uniform(0, span, N))
    if rng.uniform() < 0.25:
        t = rng.permutation(t)
    recs, edges, shifts = [], [], []
    for d in range(D):
        period = 10 ** rng.uniform(0.3, 1.3, Pn)
        t0 = rng.uniform(0, period)
        ttvs, inds = [], []
        for p in range(Pn):
            n_tr = int(min(40, max(1, (span * rng.uniform(0.3, 1.2) - t0[p]) // period[p] + 1)))
            keep = np.sort(rng.choice(n_tr, size=max(1, int(n_tr * rng.uniform(0.5, 1.0))), replace=False))
            keep[0] = 0 if rng.uniform() < 0.5 else keep[0]
            inds.append(keep)
            ttvs.append(10 ** rng.uniform(-3, -0.7) * period[p] * rng.normal(size=keep.size))
        ecc = np.where(rng.uniform(size=Pn) < 0.3, 0.0, rng.uniform(0, 0.6, Pn))
        orbit = P.TTVOrbit(period=period, t0=t0, b=rng.uniform(0, 0.9, Pn), ecc=ecc, omega=rng.uniform(-np.pi, np.pi, Pn),
                           ttvs=ttvs, transit_inds=inds)
        rr = 10 ** rng.uniform(-2, -0.7, Pn)
        try:
            recs.append(make_record(orbit, rr, sbr=0.3, window=window)[0])
        except AssertionError:
            rec = make_record(orbit, rr, sbr=0.3, window=False)[0]
            recs.append(rec)
        e, s = orbit.kernel_tables()
        if arbitrary:
            e = np.sort(rng.uniform(-0.1 * span, 1.1 * span, e.shape), axis=1)
            s = rng.uniform(-1, 1, s.shape) * period[:, None]
        edges.append(e); shifts.append(s)
    E = max(e.shape[1] for e in edges)
    edges = np.stack([np.pad(e, ((0, 0), (0, E - e.shape[1])), constant_values=np.inf) for e in edges])
    shifts = np.stack([np.pad(x, ((0, 0), (0, E + 1 - x.shape[1])), mode="edge") for x in shifts])
    rec = np.stack(recs)
    c = np.repeat(np.concatenate([P.get_cl(0.3, 0.2), P.get_cl(0.4, 0.1)])[None], D, 0)
    c = c if secondary else c[:, :3]
    g = rng.normal(size=(D, N, Pn) if per_planet else (D, N))
    kw, ckw = {}, {}
    if use_texp:
        sdt, sw = P.exposure_stencil(int(rng.choice([3, 5, 7])), int(rng.integers(0, 3)))
        te = 10 ** rng.uniform(-2.5, 0.0)      # up to a day: crosses bin edges
        if rng.uniform() < 0.3:                 # one exposure time per cadence
            te = te * rng.uniform(0.2, 1.0, N)
        kw = dict(texp=T(np.atleast_1d(te)), stencil_dt=T(sdt), stencil_w=T(sw))
        ckw = dict(texp=te, stencil_dt=sdt, stencil_w=sw)
    flags = (ops.FLAG_SECONDARY if secondary else 0) | (ops.FLAG_PER_PLANET if per_planet else 0) | (ops.FLAG_WINDOW if window else 0)
    f, gp, gl, gs = ops.transit_flux_value_and_vjp(T(t), T(rec), T(c), T(g), flags=flags, ttv=(T(edges), T(shifts)), **kw)
    wf, wgp, wgl, wgs = P.transit_flux_vjp(t, rec, c, g, per_planet=per_planet, window=window, secondary=secondary,
                                           ttv=(edges, shifts), **ckw)
    ef = float(np.abs(f.cpu().numpy() - wf).max())
    sc = np.abs(wgp).max(axis=(0, 1), keepdims=True) + 1e-300
    egp = float((np.abs(gp.cpu().numpy() - wgp) / sc).max())
    egl = float(np.abs(gl.cpu().numpy() - wgl).max() / (np.abs(wgl).max() + 1e-300))
    egs = float(np.abs(gs.cpu().numpy() - wgs).max() / (np.abs(wgs).max() + 1e-300))
    for k, v in (("flux", ef), ("gp", egp), ("gl", egl), ("gs", egs)):
        worst[k] = max(worst[k], v)
    if ef > 1e-12 or egp > 1e-8 or egl > 1e-8 or egs > 1e-8:
        print(f"case {case}: D={D} P={Pn} N={N} sec={secondary} pp={per_planet} win={window} texp={use_texp} arb={arbitrary}: "
              f"flux {ef:.1e} gparams {egp:.1e} gld {egl:.1e} gshift {egs:.1e}")
print("worst |dflux| %.2e, gparams rel %.2e, gld rel %.2e, gshift rel %.2e over %d cases"
      % (worst["flux"], worst["gp"], worst["gl"], worst["gs"], n_cases))
