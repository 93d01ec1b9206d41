#!/usr/bin/env python3
"""Time of the matrix KNN kernel against k with lowest-index ties (no restatement): is there a cliff between 16 and 17?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpeflow_amd.csrc import wrapper as W
dev = torch.device("cuda", 0)
def timed(f, iters=40):
    for _ in range(8):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
g = torch.Generator().manual_seed(0)
for B, M, Q in [(8, 8192, 4096), (4, 4096, 4096)]:
    cloud = torch.cat([torch.rand(B, 1, M, generator=g) * 29 - 14.5, torch.rand(B, 1, M, generator=g) * 17 - 8.5, torch.rand(B, 1, M, generator=g) * 91 + 22], 1).to(dev)
    query = cloud[:, :, :Q]
    print("%dx%d->%d:" % (B, M, Q), " ".join("k=%d %.1f" % (k, timed(lambda: W.k_nearest_neighbor_ties(cloud, query, k, algo="sweep", ties="index"))) for k in (8, 12, 14, 15, 16, 17, 18, 20, 24)), flush=True)
