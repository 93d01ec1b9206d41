#!/usr/bin/env python3
"""Per-level timing of the PointConv layers at the forward's shapes (B=4): pack pass + fused kernel, HIP events, graph replay."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rpeflow_amd import pointconv as PC  # noqa: E402
from rpeflow_amd.csrc import k_nearest_neighbor  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):  # ten launches per replay: the ~10 us replay overhead is amortised, dispatch gaps stay in
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters / 10 * 1e3


def main():
    torch.manual_seed(0)
    print("%-46s %9s %9s %9s %8s" % ("layer", "pack us", "fused us", "GFLOP", "TFLOP/s"))
    for N in (4096, 2048, 1024, 512, 256):  # FlowEstimator3D: conv1 195 -> 128, conv2 128 -> 128, B = 4
        xyz = torch.randn(4, 3, N, device=dev)
        knn = k_nearest_neighbor(xyz, xyz, 16)
        for C, Cout in ((195, 128), (128, 128)):
            m = PC.PointConvNoSampling(C, Cout).to(dev).eval()
            feat = torch.randn(4, C, N, device=dev)
            with torch.no_grad():
                packed = PC.pack_rows(xyz, feat)
                t_pack = timeit(lambda: PC.pack_rows(xyz, feat))
                t_fused = timeit(lambda: m(xyz, packed, knn))
            gf = 4 * N * (2 * 16 * (C + 3) * Cout + 2 * 16 * 16 * (C + 3)) / 1e9
            print("%-46s %9.1f %9.1f %9.2f %8.1f" % ("NoSampling N=%d %d->%d" % (N, C, Cout), t_pack, t_fused, gf, gf / t_fused * 1e3))
    for (M, Q, C) in ((8192, 4096, 32), (4096, 2048, 64), (2048, 1024, 96), (1024, 512, 128), (512, 256, 192)):  # FeaturePyramid3D, 2B = 8
        xyz = torch.randn(8, 3, M, device=dev)
        q = xyz[:, :, :Q]
        knn = k_nearest_neighbor(xyz, q, 16)
        m = PC.PointConvDownSampling(C, C, norm="batch_norm").to(dev).eval()
        feat = torch.randn(8, C, M, device=dev)
        with torch.no_grad():
            packed = PC.pack_rows(xyz, feat)
            t_pack = timeit(lambda: PC.pack_rows(xyz, feat))
            t_fused = timeit(lambda: m(xyz, packed, q, knn))
        gf = 8 * Q * (2 * 16 * (C + 3) * C + 2 * 16 * 16 * (C + 3)) / 1e9
        print("%-46s %9.1f %9.1f %9.2f %8.1f" % ("DownSampling %d->%d C=%d" % (M, Q, C), t_pack, t_fused, gf, gf / t_fused * 1e3))



if __name__ == "__main__":
    main()
