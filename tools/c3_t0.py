"""C3 step (bench.workload_c3) with the reference transit time moved along the series: python tools/c3_t0.py [t0 ...]
The draws' transits drift apart away from t0 (their periods differ); where along the series that happens decides which
chunks' waves are the slow ones on the sparse-mean route (exo_celerite.hip, chunk_of_block).  Prints ms per step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for t0 in [float(x) for x in sys.argv[1:]] or [1.0, 104.0, 207.0]:
    wl = bench.workload_c3(xo, ops, dev, 1024)
    with torch.no_grad():
        wl.leaves[wl.names.index("t0")].add_(t0 - 1.0)
    g = xo.GraphedStep(wl.fn, *wl.leaves)
    q = bench.time_events(lambda: g(), dev, 40)
    print("t0=%.1f sparse_mean=%s median_ms=%.4f" % (t0, bench.GP_MEAN_SPARSE, q["median_ms"]), flush=True)
    ops.release_sorted(wl.data["t"])
    del g, wl
    torch.cuda.empty_cache()
