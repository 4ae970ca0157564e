"""torch / library kernels of one eager bench step, by name and count: python tools/which_ops.py [c2|c3|c4|c5] [draws]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = bench.WORKLOADS[which](xo, ops, dev, int(sys.argv[2]) if len(sys.argv) > 2 else 256)
for _ in range(3):
    wl.fn(*wl.leaves)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    wl.fn(*wl.leaves)
    torch.cuda.synchronize()
for e in prof.key_averages():
    if e.device_type == torch.autograd.DeviceType.CUDA or "aten::" in e.key:
        print("%-70s x%d" % (e.key[:70], e.count))
