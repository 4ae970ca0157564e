"""Radial-velocity leg of a joint fit: 1024 draws x 2 planets x 500 epochs, value + gradient of
every orbit parameter, through the fused op (KeplerianOrbit.get_radial_velocity) and through the
composed formulas (ops.kepler + torch), eager and as a hipGraph.   python tools/profile_rv.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import exoplanet_amd as xo

dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
D, N = 1024, 500
T = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev).requires_grad_(True)
leaves = dict(period=T(10 ** rng.uniform(0.5, 2, (D, 2))), t0=T(rng.uniform(0, 5, (D, 2))), ecc=T(rng.uniform(0, 0.6, (D, 2))),
              omega=T(rng.uniform(-3, 3, (D, 2))), K=T(rng.uniform(1, 20, (D, 2))))
b_fixed = torch.as_tensor(rng.uniform(0, 0.5, (D, 2)), dtype=torch.float64, device=dev)   # no role in K-parameterised RV
t = torch.linspace(0, 300, N, dtype=torch.float64, device=dev)
y = torch.randn(N, dtype=torch.float64, device=dev)
names = list(leaves)


def make(route):
    def step(*vals):
        L = dict(zip(names, vals))
        orbit = xo.KeplerianOrbit(period=L["period"], t0=L["t0"], ecc=L["ecc"], omega=L["omega"], b=b_fixed)
        if route == "fused":
            rv = orbit.get_radial_velocity(t, K=L["K"])
        else:
            sinf, cosf = orbit._get_true_anomaly(t)                                   # (D, N, 2)
            cw, sw, e = (x.unsqueeze(-2) for x in (orbit.cos_omega, orbit.sin_omega, orbit.ecc))
            rv = L["K"].unsqueeze(-2) * (cw * cosf - sw * sinf + e * cw)             # keplerian.py:660-669
        chi2 = ((y[None, :] - rv.sum(-1)) ** 2).sum()
        return (chi2.detach(),) + torch.autograd.grad(chi2, vals)
    return step


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


out = {"draws": D, "planets": 2, "epochs": N}
vals = tuple(leaves.values())
ref = None
for route in ("fused", "composed"):
    step = make(route)
    res = step(*vals)
    ref = res if ref is None else ref
    g = xo.GraphedStep(step, *vals)
    out[route] = {"eager_ms": timeit(lambda: step(*vals)), "graph_ms": timeit(lambda: g()),
                  "max_rel_grad_difference_vs_fused": max(float((a - b).abs().max() / (b.abs().max() + 1e-300))
                                                          for a, b in zip(res[1:], ref[1:]))}
print(json.dumps(out, indent=1))
