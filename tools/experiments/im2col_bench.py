#!/usr/bin/env python3
"""rpe_im2col at the context network's shapes: us per launch and GB/s of stores."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpeflow_amd import utils as U
dev = torch.device("cuda", 0)
for B, C, H, W, d in [(4, 128, 72, 120, 2), (4, 128, 72, 120, 16), (4, 128, 36, 60, 4), (4, 128, 18, 30, 2), (4, 430, 9, 15, 1), (8, 64, 36, 60, 1)]:
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(128, C, 3, 3, device=dev)
    f = lambda: U.im2col_conv(x, w, None, (1, 1), (d, d), (d, d))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=d, dilation=d)
    err = float((f().double() - ref).abs().max() / ref.abs().max())
    import rpeflow_amd._lib as L
    cols = torch.empty(B, C * 9, H * W, device=dev)
    g = lambda: L.check(L.lib().rpe_im2col(U._ptr(x), B, C, H, W, 3, 3, 1, 1, d, d, d, d, U._ptr(cols), L.stream_of(x)), "im2col")
    for _ in range(5): g()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): g()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 50 * 1e3
    print("B%d C%d %dx%d dil %d: im2col %.1f us (%.2f TB/s of stores), conv rel err %.1e" % (B, C, H, W, d, us, cols.numel() * 4 / us / 1e6, err))
