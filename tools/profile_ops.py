"""the reference's standalone Ops at n = 1.5e8 (bench.py extras.ops) on their own, for rocprofv3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from exoplanet_amd import ops
print(bench.extra_ops(ops, torch.device("cuda:0")))
