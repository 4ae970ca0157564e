"""User-level leapfrog step with a TTVOrbit (C2 shape: 150 000 cadences, 1024 draws, 60 labelled
transits, every draw its own offsets): leaves -> TTVOrbit -> LimbDarkLightCurve.get_light_curve
-> sum(gbar * flux) -> gradients of every leaf (the 60 offsets per draw included); issued eagerly
and replayed as one hipGraph.   python tools/ttv_step.py [draws]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import exoplanet_amd as xo
from exoplanet_amd.orbits import TTVOrbit

dev = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(2)
t = torch.arange(150_000, dtype=torch.float64, device=dev) * (2.0 / 1440.0)
n_tr = 60
base = dict(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1, r=0.1)
leaves = {k: torch.tensor(v * (1 + 1e-3 * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev, requires_grad=True)
          for k, v in base.items()}
leaves["u1"] = torch.tensor(0.3 * (1 + 1e-3 * rng.normal(size=D)), dtype=torch.float64, device=dev, requires_grad=True)
leaves["u2"] = torch.tensor(0.2 * (1 + 1e-3 * rng.normal(size=D)), dtype=torch.float64, device=dev, requires_grad=True)
leaves["ttvs"] = torch.tensor(0.01 * rng.normal(size=(D, n_tr)), dtype=torch.float64, device=dev, requires_grad=True)
gbar = torch.randn(D, t.numel(), 1, dtype=torch.float64, device=dev)
names = list(leaves)


def step(*vals):
    L = dict(zip(names, vals))
    orbit = TTVOrbit(period=L["period"], t0=L["t0"], b=L["b"], ecc=L["ecc"], omega=L["omega"], ttvs=[L["ttvs"]])
    flux = xo.LimbDarkLightCurve(L["u1"], L["u2"]).get_light_curve(orbit=orbit, r=L["r"], t=t, use_in_transit=False)
    loss = (flux * gbar).sum()
    return (loss.detach(),) + torch.autograd.grad(loss, vals)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def step_dot(*vals):
    """the same step with the cotangent handed to the kernels (bench.py's form): no (D, N) torch passes"""
    from exoplanet_amd import ops

    L = dict(zip(names, vals))
    orbit = TTVOrbit(period=L["period"], t0=L["t0"], b=L["b"], ecc=L["ecc"], omega=L["omega"], ttvs=[L["ttvs"]])
    rec, ld, _, flags = orbit.kernel_inputs(L["r"], (L["u1"], L["u2"]))
    edges, shift = orbit.kernel_ttv()
    flux, dot = ops.transit_flux_dot(t, rec, ld, gbar[..., 0], flags=flags, ttv=(edges.contiguous(), shift.contiguous()))
    return (dot.sum().detach(),) + torch.autograd.grad(dot.sum(), vals)


vals = tuple(leaves.values())
out = {"draws": D, "n_cad": t.numel(), "transits": n_tr}
for label, fn in (("get_light_curve_then_torch_loss", step), ("kernel_inputs_then_transit_flux_dot", step_dot)):
    eager = timeit(lambda: fn(*vals))
    g = xo.GraphedStep(fn, *vals)
    graphed = timeit(lambda: g())
    out_e, out_g = fn(*vals), g()
    err = max(float((a - b).abs().max() / (b.abs().max() + 1e-300)) for a, b in zip(out_e[1:], out_g[1:]))
    out[label] = {"eager_ms": eager * 1e3, "graph_ms": graphed * 1e3, "evals_per_s_graph": D / graphed,
                  "max_rel_grad_difference_eager_vs_graph": err}
a, b = step(*vals), step_dot(*vals)
out["max_rel_grad_difference_between_the_two_forms"] = max(float((x - y).abs().max() / (y.abs().max() + 1e-300))
                                                           for x, y in zip(a[1:], b[1:]))
print(json.dumps(out, indent=1))
