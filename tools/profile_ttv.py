"""Cost of transit-timing variations in the fused kernels: BASELINE config C2 (150 000 cadences,
1024 draws, value + gradient) without and with per-draw timing tables (60 labelled transits).
    python tools/profile_ttv.py > gpurun_out/ttv.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from exoplanet_amd import ops  # noqa: E402
from oracle import numpy_port as P  # noqa: E402
from test_gpu_transit import make_record  # noqa: E402
from tools.bench_configs import perturb, timeit  # noqa: E402

dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)  # noqa: E731


def main():
    rng = np.random.default_rng(1)
    t = np.arange(150_000) * (2.0 / 1440.0)
    D = int(os.environ.get("TTV_DRAWS", 1024))
    period, t0 = 3.5, 1.0
    n_tr = int((t[-1] - t0) / period) + 1
    out = {"n_cad": t.size, "draws": D, "transits": n_tr}
    for label, texp in (("no_texp", None), ("texp_7", 0.02)):
        kw = {}
        if texp is not None:
            sdt, sw = P.exposure_stencil(7, 0)
            kw = dict(texp=T([texp]), stencil_dt=T(sdt), stencil_w=T(sw))
        recs, edges, shifts = [], [], []
        base = None
        for d in range(D):
            orbit = P.TTVOrbit(period=np.array([period]), t0=np.array([t0]), b=np.array([0.3]), ecc=np.array([0.3]),
                               omega=np.array([1.1]), ttvs=[0.01 * rng.normal(size=n_tr)])
            if base is None:
                base = make_record(orbit, np.array([0.1]))
            e, s = orbit.kernel_tables()
            edges.append(e)
            shifts.append(s)
        recs = perturb(base, D, rng)
        c = np.repeat(P.get_cl(0.3, 0.2)[None], D, 0)
        tt, rt, ct = T(t), T(recs), T(c)
        gt = torch.randn(D, t.size, dtype=torch.float64, device=dev)
        ttv = (T(np.stack(edges)), T(np.stack(shifts)))
        if os.environ.get("TTV_ONE_BIN"):     # how much do the per-tile checks cost when nothing is ever looked up?
            ttv = (torch.full_like(ttv[0], float("inf")), torch.zeros_like(ttv[1]))
        plain = timeit(lambda: ops.transit_flux_value_and_vjp(tt, rt, ct, gt, **kw), 10)
        with_ttv = timeit(lambda: ops.transit_flux_value_and_vjp(tt, rt, ct, gt, ttv=ttv, **kw), 10)
        fwd_plain = timeit(lambda: ops.transit_flux(tt, rt, ct, **kw), 10)
        fwd_ttv = timeit(lambda: ops.transit_flux(tt, rt, ct, ttv=ttv, **kw), 10)
        out[label] = {"value_grad_ms": {"keplerian": plain * 1e3, "ttv": with_ttv * 1e3},
                      "value_ms": {"keplerian": fwd_plain * 1e3, "ttv": fwd_ttv * 1e3},
                      "ttv_evals_per_s": D / with_ttv}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
