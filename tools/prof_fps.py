"""FPS only (2B=8 clouds of 8192 -> 4096), for rocprofv3 runs.  Usage: python tools/prof_fps.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rpeflow_amd.csrc as ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.manual_seed(0)
p = (torch.rand(8, 3, 8192, device="cuda:0") * 30).transpose(1, 2)
for _ in range(iters):
    ops.furthest_point_sampling(p, 4096)
torch.cuda.synchronize()
print("done", iters)
