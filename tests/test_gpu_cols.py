"""GPU: the whole user-level step in one call of the library (round 5; exo_transit_flux_cols_vjp_f64): the constructor's
columns in, flux, per-draw scalar and COLUMN gradients out -- the packing riding on the windows + enumeration launch, the
packing VJP on the sweep's last kernel -- against the same step through autograd (ops.orbit_flux_dot + torch.autograd.grad:
packing kernel, sweep, packing-VJP kernel).  The same device functions on the same numbers: bit-equal."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    return torch.device("cuda:0")


def _leaves(dev, D, P, seed, circular=False):
    rng = np.random.default_rng(seed)
    base = dict(period=[3.5, 7.9, 13.1, 29.7][:P], t0=[1.0, 2.3, 5.1, 11.7][:P], b=[0.3, 0.1, 0.5, 0.2][:P],
                ecc=[0.3, 0.1, 0.2, 0.05][:P], omega=[1.1, -0.4, 2.0, 0.3][:P])
    if circular:
        del base["ecc"], base["omega"]
    L = {k: torch.tensor(np.asarray(v)[None] * (1 + 1e-3 * rng.normal(size=(D, P))), dtype=torch.float64, device=dev,
                         requires_grad=True) for k, v in base.items()}
    r = torch.tensor(np.asarray([0.1, 0.05, 0.07, 0.03][:P])[None] * (1 + 1e-3 * rng.normal(size=(D, P))), dtype=torch.float64,
                     device=dev, requires_grad=True)
    u1 = torch.tensor(0.3 * (1 + 0.01 * rng.normal(size=D)), dtype=torch.float64, device=dev, requires_grad=True)
    u2 = torch.tensor(0.2 * (1 + 0.01 * rng.normal(size=D)), dtype=torch.float64, device=dev, requires_grad=True)
    return L, r, u1, u2


CASES = [
    dict(D=1024, P=1, N=30_000),                                    # a block finishes its own draw: two launches
    dict(D=64, P=4, N=20_000),                                      # C4's shape: the finish kernel carries the packing VJP
    dict(D=70, P=2, N=9_001, texp=0.02),                            # exposure stencil
    dict(D=33, P=1, N=12_000, secondary=True, texp=29.4 / 1440, cadence=29.4 / 1440),   # transits + occultations
    dict(D=40, P=1, N=8_000, use_in_transit=True),                  # the reference's windows from the packing
    dict(D=20, P=2, N=6_000, circular=True, extra="m_star"),        # ecc=None; a broadcast (P,) column with a gradient
    dict(D=16, P=1, N=5_000, sparse=True),
    dict(D=12, P=1, N=4_000, unsorted=True),                        # not a fused launch: the three calls one after the other
    dict(D=600, P=1, N=10_000, gscale=True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k in ("D", "P")) + "-" + "-".join(
    k for k in c if k not in ("D", "P", "N", "texp", "cadence")))
def test_value_and_grad_equals_autograd(dev, case):
    import exoplanet_amd as xo
    from exoplanet_amd import ops
    from exoplanet_amd.light_curves.limb_dark import exposure_stencil, _on_device

    D, P, N = case["D"], case["P"], case["N"]
    t = torch.arange(N, dtype=torch.float64, device=dev) * case.get("cadence", 2.0 / 1440.0)
    if case.get("unsorted"):
        t = t.flip(0).contiguous()
    L, r, u1, u2 = _leaves(dev, D, P, 5, circular=case.get("circular", False))
    extra = {}
    if case.get("extra") == "m_star":
        extra["m_star"] = torch.tensor([1.1], dtype=torch.float64, device=dev, requires_grad=True)
        extra["r_star"] = torch.tensor([0.9], dtype=torch.float64, device=dev, requires_grad=True)
    g = torch.randn(D, N, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    kw = dict(use_in_transit=case.get("use_in_transit", False), sparse=case.get("sparse", False))
    if "texp" in case:
        dt, w = exposure_stencil(7, 0)
        kw.update(texp=torch.tensor([case["texp"]], dtype=torch.float64, device=dev), stencil=(_on_device(dt, dev), _on_device(w, dev)))
    sec = None
    if case.get("secondary"):
        sbr = torch.full((D,), 0.3, dtype=torch.float64, device=dev, requires_grad=True)
        us = (torch.full((D,), 0.4, dtype=torch.float64, device=dev, requires_grad=True),
              torch.full((D,), 0.1, dtype=torch.float64, device=dev, requires_grad=True))
        sec = (us, sbr)
        kw["secondary"] = sec
    gscale = torch.linspace(0.5, 1.5, D, dtype=torch.float64, device=dev) if case.get("gscale") else None
    orbit = xo.KeplerianOrbit(**L, **extra)
    flux_a, L_a = orbit.flux_dot(r, (u1, u2), t, g, **kw)
    leaves = dict(L)
    leaves.update(r=r, u1=u1, u2=u2, **extra)
    if sec is not None:
        leaves.update(sbr=sec[1], u1s=sec[0][0], u2s=sec[0][1])
    ga = torch.autograd.grad((L_a * gscale).sum() if gscale is not None else L_a.sum(), list(leaves.values()))
    flux_b, L_b, gb = xo.KeplerianOrbit(**L, **extra).flux_value_and_grad(r, (u1, u2), t, g, gscale=gscale, **kw)
    torch.cuda.synchronize()
    if case.get("sparse"):
        assert torch.equal(flux_a.vals[:, :, :10], flux_b.vals[:, :, :10]) and flux_b.n_solved() == flux_a.n_solved() > 0
    else:
        assert torch.equal(flux_a, flux_b) and bool((flux_b != 0).any())
    assert torch.equal(L_a.detach(), L_b)
    assert set(gb) == set(leaves), (sorted(gb), sorted(leaves))
    for (name, x), a in zip(leaves.items(), ga):
        assert gb[name].shape == x.shape
        assert torch.equal(a, gb[name]), (name, float((a - gb[name]).abs().max()), float(a.abs().max()))
        assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0, name


def test_value_and_grad_replayed_as_a_hip_graph(dev):
    """captured and replayed with changing leaves: what the bench's C2 step does"""
    import exoplanet_amd as xo
    from exoplanet_amd import ops

    D, N = 1024, 20_000
    t = ops.vouch_sorted(torch.arange(N, dtype=torch.float64, device=dev) * (2.0 / 1440.0))
    L, r, u1, u2 = _leaves(dev, D, 1, 9)
    names = list(L)
    g = torch.randn(D, N, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    static = [x.detach().clone() for x in list(L.values()) + [r, u1, u2]]

    def step(*v):
        flux, Lv, grads = xo.KeplerianOrbit(**dict(zip(names, v[:5]))).flux_value_and_grad(v[5], (v[6], v[7]), t, g)
        return (flux, Lv) + tuple(grads[k] for k in names + ["r", "u1", "u2"])

    graph = xo.GraphedStep(step, *static)
    out = [x.clone() for x in graph()]
    eager = step(*static)
    for a, b in zip(out, eager):
        assert torch.equal(a, b)
    with torch.no_grad():
        static[0].mul_(1.0002)
    out2 = graph()
    eager2 = step(*static)
    for a, b in zip(out2, eager2):
        assert torch.equal(a, b)
    assert not torch.equal(out[1], out2[1])
    ops.release_sorted(t)
