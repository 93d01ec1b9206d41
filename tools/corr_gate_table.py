#!/usr/bin/env python3
"""Which correlation kernel takes which map?  Every kernel of csrc/correlation.hip (algo 1 direct, 2 register-staged MFMA, 3 small-map,
7 / 8 LDS-DMA ring with two rows / one row a wave; profiles/r06_corr_gate_table_before.txt also lists the four-wave forms 4 / 9 that
left the library) on every correlation2d call of the forward (RPEFlow_core.py:362; B = 4 and the DSEC batch of 3),
alone, 200 launches between two events; "auto" is what rpe_correlation2d_forward's gate picks.  Results are compared with the
direct kernel's (max |diff|)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rpeflow_amd.csrc import wrapper as W  # noqa: E402

DEV = "cuda:0"
SHAPES = [("things L1", 4, 32, 144, 240), ("things L2", 4, 64, 72, 120), ("things L3", 4, 96, 36, 60), ("things L4", 4, 128, 18, 30),
          ("things L5", 4, 192, 9, 15), ("dsec L1", 3, 32, 128, 160), ("dsec L2", 3, 64, 64, 80), ("dsec L3", 3, 96, 32, 40),
          # other batch sizes around the one-row / two-row break-even (1024 waves of the two-row tiling), and the wide config-2 map cropped
          ("B1 L1", 1, 32, 144, 240), ("B2 L1", 2, 32, 144, 240), ("B3 L1", 3, 32, 144, 240), ("B8 L1", 8, 32, 144, 240),
          ("B8 L2", 8, 64, 72, 120), ("B16 L2", 16, 64, 72, 120), ("B1 288x480", 1, 16, 288, 480), ("B1 C256 272x480", 1, 256, 272, 480)]


def timed(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    print("%-12s %3s %4s %4s %4s | %s | auto" % ("call", "B", "C", "H", "W", " ".join("%9s" % ("algo %d" % a) for a in (1, 2, 3, 7, 8))))
    g = torch.Generator().manual_seed(0)
    for name, B, C, H, Wd in SHAPES:
        a, b = torch.randn(B, C, H, Wd, generator=g).to(DEV), torch.randn(B, C, H, Wd, generator=g).to(DEV)
        ref = W._correlation2d_algo(a, b, 4, 1, leaky_slope=0.1)
        cells = []
        for algo in (1, 2, 3, 7, 8):
            try:
                out = W._correlation2d_algo(a, b, 4, algo, leaky_slope=0.1)
                err = (out - ref).abs().max().item()
                us = timed(lambda: W._correlation2d_algo(a, b, 4, algo, leaky_slope=0.1))
                cells.append("%7.1f%s" % (us, " " if err < 5e-6 else "!"))
            except RuntimeError:
                cells.append("      - ")
        auto = timed(lambda: W._correlation2d_algo(a, b, 4, 0, leaky_slope=0.1))
        print("%-12s %3d %4d %4d %4d | %s | %6.1f us" % (name, B, C, H, Wd, "  ".join(cells), auto), flush=True)


if __name__ == "__main__":
    main()
