"""Correlation microbench only (BASELINE config 2), for rocprofv3 --pmc / --kernel-trace runs.
Usage: python tools/prof_corr.py [algo] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rpeflow_amd.csrc import wrapper as W

algo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
a = torch.randn(1, 256, 544, 960, device="cuda:0")
b = torch.randn(1, 256, 544, 960, device="cuda:0")
for _ in range(iters):
    W._correlation2d_algo(a, b, 4, algo)
torch.cuda.synchronize()
print("done", algo, iters)
