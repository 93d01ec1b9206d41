#!/usr/bin/env python3
"""What can the memory system take for the level-1 project_in layer (96 -> 510 channels over 4 x 144 x 240: 53 MB in, 282 MB out)?
Times a fill of the output tensor, a copy of the same size, and the read of the input, with HIP events over 100 launches:
the layer's floors beside its matrix time (13.5 GFLOP at 157.3 TF = 86 us)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

dev = torch.device("cuda", 0)


def timed(fn, iters=100):
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


y = torch.empty(4, 510, 144, 240, device=dev)
y2 = torch.empty_like(y)
x = torch.randn(4, 96, 144, 240, device=dev)
mb = lambda t: t.numel() * 4 / 1e6
t_fill = timed(lambda: y.fill_(1.0))
t_copy = timed(lambda: y2.copy_(y))
t_read = timed(lambda: x.sum())
print("fill  %6.1f MB: %6.1f us  %5.2f TB/s written" % (mb(y), t_fill, mb(y) / t_fill))
print("copy  %6.1f MB: %6.1f us  %5.2f TB/s read + written" % (mb(y), t_copy, 2 * mb(y) / t_copy))
print("read  %6.1f MB: %6.1f us  %5.2f TB/s (a reduction over the input)" % (mb(x), t_read, mb(x) / t_read))
