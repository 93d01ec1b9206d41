#!/usr/bin/env python3
"""What is the ~14.5 us of a tiny neighbour search (4 x (256 -> 256)) made of?  The same search with the tie restatement off
(ties="index"), on a cloud without ties (unit cube), with one sample, and an empty kernel's launch time for scale."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from rpeflow_amd.csrc.wrapper import k_nearest_neighbor_ties

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def time_us(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


ids = (torch.rand(4, 3, 512, generator=g) * torch.tensor([29.0, 17.0, 91.0])[None, :, None] + torch.tensor([-14.5, -8.5, 22.0])[None, :, None]).to(dev)
unit = torch.rand(4, 3, 512, generator=g).to(dev)
x = torch.zeros(64, device=dev)
print("empty-ish kernel (x.add_(1) on 64 floats): %.1f us" % time_us(lambda: x.add_(1.0)))
for name, c in (("IDS range", ids), ("unit cube", unit)):
    for n in (256, 512):
        for k in (16, 3):
            p = c[:, :, :n].contiguous()
            for ties in ("torch", "index"):
                t = time_us(lambda: k_nearest_neighbor_ties(p, p, k, ties=ties))
                t1 = time_us(lambda: k_nearest_neighbor_ties(p[:1], p[:1], k, ties=ties))
                print("%-10s n=%4d k=%2d ties=%-5s  B=4: %5.1f us   B=1: %5.1f us" % (name, n, k, ties, t, t1))
