#!/usr/bin/env python3
"""The level-1 / level-2 3x3 convolutions of the flow estimator and the context network through MIOpen in NCHW and in
channels-last: time per call (which solver family wins: Winograd or the NHWC implicit GEMM) and run-to-run reproducibility."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)
def timed(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (B, cin, cout, H, W, d) in [(4, 243, 128, 144, 240, 1), (4, 128, 128, 144, 240, 1), (4, 128, 96, 144, 240, 1), (4, 96, 64, 144, 240, 1), (4, 64, 32, 144, 240, 1),
                                (4, 128, 128, 144, 240, 2), (4, 243, 128, 72, 120, 1), (4, 128, 128, 72, 120, 1), (4, 34, 128, 144, 240, 1), (4, 96, 2, 144, 240, 1)]:
    x = torch.randn(B, cin, H, W, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev)
    xl, wl = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
    t1 = timed(lambda: F.conv2d(x, w, padding=d, dilation=d))
    t2 = timed(lambda: F.conv2d(xl, wl, padding=d, dilation=d))
    outs = [F.conv2d(xl, wl, padding=d, dilation=d).clone() for _ in range(8)]
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    diff = float((outs[0] - F.conv2d(x, w, padding=d, dilation=d)).abs().max())
    gf = 2.0 * cin * cout * 9 * B * H * W / 1e9
    print("B%d %3d->%3d %dx%d dil %d (%.1f GF): NCHW %.0f us (%.0f TF) | channels-last %.0f us (%.0f TF) reproducible %s, max diff to NCHW %.1e" % (
        B, cin, cout, H, W, d, gf, t1, gf / t1 * 1e3, t2, gf / t2 * 1e3, same, diff), flush=True)
