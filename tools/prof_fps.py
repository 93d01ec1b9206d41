"""FPS only (2B=8 clouds of 8192 -> 4096), for rocprofv3 runs.  Usage: python tools/prof_fps.py [variant] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rpeflow_amd.csrc as ops
from rpeflow_amd import _lib

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
_lib.lib().rpe_debug_set_fps_variant(variant)
torch.manual_seed(0)
p = (torch.rand(8, 3, 8192, device="cuda:0") * 30).transpose(1, 2)
for _ in range(iters):
    ops.furthest_point_sampling(p, 4096)
torch.cuda.synchronize()
print("done", variant, iters)
