#!/usr/bin/env python
"""GB/s of the standalone Ops at n elements for the library EXOPLANET_AMD_LIB selects: python tools/ops_bench.py [n] [iters]
(kepler 32 B / element; quad_solution_vector 40 B value, 88 B with derivatives) -> one JSON line"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exoplanet_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(11)


def timed(fn):
    for _ in range(2):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ms[len(ms) // 2]


out = {"lib": os.environ.get("EXOPLANET_AMD_LIB", "product"), "n": n}
with torch.no_grad():
    M = (torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5) * 800.0
    e = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 0.9
    ms = timed(lambda: ops.kepler(M, e))
    out["kepler_ms"], out["kepler_GBps"] = round(ms, 4), round(32.0 * n / ms / 1e6, 1)
    e0 = torch.zeros_like(e)
    ms = timed(lambda: ops.kepler(M, e0))
    out["kepler_circular_GBps"] = round(32.0 * n / ms / 1e6, 1)
    # a plain copy of the same bytes (two arrays in, two out), torch's elementwise kernel: what the memory side allows
    s, c = torch.empty_like(M), torch.empty_like(M)
    ms = timed(lambda: (s.copy_(M), c.copy_(e)))
    out["copy_2in_2out_GBps"] = round(32.0 * n / ms / 1e6, 1)
    del M, e, e0, s, c
    b = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 1.3
    r = torch.full((n,), 0.1, dtype=torch.float64, device=dev)
    ms = timed(lambda: ops.quad_solution_vector(b, r))
    out["quad_sv_GBps"] = round(40.0 * n / ms / 1e6, 1)
    ms = timed(lambda: ops.quad_solution_vector_derivs(b, r))
    out["quad_sv_grad_GBps"] = round(88.0 * n / ms / 1e6, 1)
    b2 = 1.2 + torch.rand(n, dtype=torch.float64, device=dev, generator=g)       # out of transit: the majority of a light curve
    ms = timed(lambda: ops.quad_solution_vector(b2, r))
    out["quad_sv_out_of_transit_GBps"] = round(40.0 * n / ms / 1e6, 1)
print(json.dumps(out))
