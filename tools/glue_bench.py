"""Hot-path glue operators at the level-1 / level-2 shapes of the forward (B = 4), replayed from HIP graphs: kernel time
and algorithmic bytes per SURVEY.md section 8(d) -> achieved GB/s.  Usage: python tools/glue_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rpeflow_amd.csrc as ops
from rpeflow_amd import utils as U
from rpeflow_amd.pointconv import PointConvNoSampling
from rpeflow_amd.pwc3d_core import Correlation3D

dev = "cuda:0"


def graph_time(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * reps) * 1e3


def report(name, us, byts):
    print(f"{name:58s} {us:8.1f} us  {byts / 1e6:8.1f} MB  {byts / us / 1e3:7.1f} GB/s  ({byts / us / 1e3 / 8000:.3f} of 8 TB/s)")


torch.manual_seed(0)
B = 4
for (H, W, N, C2, C3) in [(144, 240, 4096, 32, 32), (144, 240, 4096, 81, 37), (72, 120, 2048, 64, 64)]:
    HW = H * W
    feat2d, feat3d = torch.randn(B, C2, H, W, device=dev), torch.randn(B, C3, N, device=dev)
    xy = torch.stack([torch.rand(B, N, device=dev) * W, torch.rand(B, N, device=dev) * H], 1)
    grid = U.mesh_grid(B, H, W, dev).reshape(B, 2, -1)
    nn = ops.k_nearest_neighbor(xy, grid, k=1)[..., 0]
    report(f"project_feat_with_nn_corr {H}x{W} N={N} C2={C2} C3={C3}", graph_time(lambda: U.project_feat_with_nn_corr(xy, feat2d, feat3d, nn)),
           4 * B * (C2 * HW + (C3 + 2) * N + (C3 + 3) * HW) + 8 * B * HW)
    report(f"grid_sample_wrapper {H}x{W} N={N} C={C2}", graph_time(lambda: U.grid_sample_wrapper(feat2d, xy)), 4 * B * (C2 * HW + 2 * N + C2 * N))
    flow = torch.randn(B, 2, H, W, device=dev) * 3
    report(f"backwarp_2d {H}x{W} C={C2}", graph_time(lambda: U.backwarp_2d(feat2d, flow, padding_mode="border")), 4 * B * HW * (2 * C2 + 2))
    report(f"knn 2-D k=1 M={N} Q={HW}", graph_time(lambda: ops.k_nearest_neighbor(xy, grid, k=1)), 4 * B * 2 * (HW + N) + 8 * B * HW)
    xyz = torch.rand(B, 3, N, device=dev) * 30
    xyzc = xyz[:, :, :N // 2].contiguous()
    featc = torch.randn(B, 67, N // 2, device=dev)
    knn3 = ops.k_nearest_neighbor(xyzc, xyz, k=3)
    report(f"knn 3-D k=3 M={N // 2} Q={N}", graph_time(lambda: ops.k_nearest_neighbor(xyzc, xyz, k=3)), 4 * B * 3 * (N + N // 2) + 8 * B * N * 3)
    report(f"knn_interpolation (with its KNN) M={N // 2} Q={N} C=67", graph_time(lambda: U.knn_interpolation(xyzc, featc, xyz, k=3)),
           4 * B * 3 * (N + N // 2) + 8 * B * N * 3 + 4 * B * (67 * N // 2 + 67 * N))
    report(f"knn 3-D k=16 M={N} Q={N}", graph_time(lambda: ops.k_nearest_neighbor(xyz, xyz, k=16)), 4 * B * 3 * 2 * N + 8 * B * N * 16)
    knn16 = ops.k_nearest_neighbor(xyz, xyz, k=16)
    pc = PointConvNoSampling(128, 128, norm=None, k=16).to(dev).eval()
    f128 = torch.randn(B, 128, N, device=dev)
    with torch.no_grad():
        report(f"PointConvNoSampling 128->128 N={N} (group + linear)", graph_time(lambda: pc(xyz, f128, knn16)),
               4 * B * 131 * N + 8 * B * N * 16 + 4 * 128 * 16 * 131 + 4 * B * 128 * N)
        c3 = Correlation3D(C3 if C3 % 2 == 0 else 32, C3 if C3 % 2 == 0 else 32, k=16).to(dev).eval()
        c = C3 if C3 % 2 == 0 else 32
        fa, fb = torch.randn(B, c, N, device=dev), torch.randn(B, c, N, device=dev)
        xyz2 = xyz + torch.randn_like(xyz) * 0.1
        report(f"Correlation3D N={N} C={c} (with its KNN)", graph_time(lambda: c3(xyz, fa, xyz2, fb, knn16)),
               4 * B * N * (2 * c + 6) + 2 * 8 * B * N * 16 + 4 * B * c * N)
    print()
