#!/usr/bin/env python3
"""The 1x1 kernel's memory pattern without its arithmetic (tools/experiments/write_pattern.hip): 96 -> 510 channels over
4 x 144 x 240, stores only and loads + stores, for the shipped tile shape (4 output tiles x 64 positions a wave: 4 planes x 256 B
per store instruction) and shapes with longer runs per plane.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC
tools/experiments/write_pattern.hip -o tools/_exp/libwrite_pattern.so (here; the box runs the .so)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_exp", "libwrite_pattern.so"))
lib.write_pattern.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
B, Cin, Cout, P = 4, 96, 510, 144 * 240
x = torch.randn(B, Cin, P, device=dev)
y = torch.empty(B, Cout, P, device=dev)


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


out_mb, in_mb = y.numel() * 4 / 1e6, x.numel() * 4 / 1e6
print("output %.0f MB, input %.0f MB; fill: %.1f us" % (out_mb, in_mb, timed(lambda: y.fill_(1.0))))
print("%-44s %10s %12s" % ("wave tile (output tiles x position groups), waves", "stores us", "loads+stores"))
for OT, PG, nw in ((4, 1, 4), (2, 1, 4), (1, 1, 4), (2, 2, 4), (2, 2, 8), (1, 4, 4), (4, 2, 4), (4, 4, 4), (4, 4, 8)):
    st = torch.cuda.current_stream().cuda_stream
    call = lambda r: lib.write_pattern(x.data_ptr(), Cin, y.data_ptr(), Cout, P, B, OT, PG, nw, r, st)
    assert call(0) == 0
    t0, t1 = timed(lambda: call(0)), timed(lambda: call(1))
    yblocks = -(-((Cout + 15) // 16) // (nw * OT))
    print("%d x %d (%3d channels x %4d positions, %4d B runs), %d waves, x read %dx: %8.1f us %10.1f us" % (
        OT, PG, 16 * OT, 64 * PG, 256 * PG, nw, yblocks, t0, t1))
