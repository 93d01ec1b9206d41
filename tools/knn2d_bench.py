#!/usr/bin/env python3
"""The model's nearest-projected-point searches (k = 1, D = 2, both frames batched: B = 8) at the five pyramid levels:
binned search (csrc/knn_binned.hip) against the sweeping kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rpeflow_amd.csrc import wrapper as W
from tests import inputs as I

dev = torch.device("cuda", 0)


def timed(f, iters=50):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


r = I.rng(0)
tot = {"binned": 0.0, "sweep": 0.0}
for B, N, H, Wd in [(8, 4096, 144, 240), (8, 2048, 72, 120), (8, 1024, 36, 60), (8, 512, 18, 30), (8, 256, 9, 15), (6, 4096, 128, 160)]:
    pts = torch.from_numpy(I.pixel_cloud(r, B, N, H, Wd)).to(dev).transpose(1, 2).contiguous()  # [B, 2, N] as the model holds them
    qry = torch.from_numpy(I.pixel_grid(B, H, Wd)).to(dev).transpose(1, 2).contiguous()
    a = W.k_nearest_neighbor_ties(pts, qry, 1, algo="binned")
    b = W.k_nearest_neighbor_ties(pts, qry, 1, algo="sweep")
    tb = timed(lambda: W.k_nearest_neighbor_ties(pts, qry, 1, algo="binned"))
    ts = timed(lambda: W.k_nearest_neighbor_ties(pts, qry, 1, algo="sweep"))
    if B == 8:
        tot["binned"] += tb
        tot["sweep"] += ts
    print("B=%d N=%5d %3dx%3d: binned %6.1f us | sweep %6.1f us | same %s" % (B, N, H, Wd, tb, ts, torch.equal(a, b)), flush=True)
print("five FlyingThings3D levels: binned %.1f us, sweep %.1f us" % (tot["binned"], tot["sweep"]))
