import json, sys, torch
sys.path.insert(0, '.')
import bench, exoplanet_amd as xo
from exoplanet_amd import ops
dev = torch.device('cuda:0')
out = bench.extra_gp_conditioning(xo, ops, dev, 1024)
for k, v in out.items():
    if isinstance(v, dict): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != 'note'})
