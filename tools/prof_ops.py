"""Single hot-path operators at the forward's level-1 shapes, for rocprofv3 --pmc / --kernel-trace runs.
Usage: python3 tools/prof_ops.py <pointconv|knn16|knn3|knn2d|corr3d|project|pw_l1> [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rpeflow_amd import pointconv as PC
from rpeflow_amd.csrc import k_nearest_neighbor

which = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
torch.manual_seed(0)
with torch.no_grad():
    if which == "pointconv":  # FlowEstimator3D.point_conv1 at level 1: B=4, N=4096, 195 -> 128 (pwc3d_core.py:123)
        xyz = torch.randn(4, 3, 4096, device=dev)
        knn = k_nearest_neighbor(xyz, xyz, 16)
        m = PC.PointConvNoSampling(195, 128).to(dev).eval()
        packed = PC.pack_rows(xyz, torch.randn(4, 195, 4096, device=dev))
        step = lambda: m(xyz, packed, knn)
    elif which in ("knn16", "knn3"):  # the forward's largest 3-D searches
        k = 16 if which == "knn16" else 3
        if k == 16:  # bench.py's roofline_knn workload exactly: 8 x (8192 -> 4096), the queries a prefix of the cloud
            g = torch.Generator(device="cpu").manual_seed(0)
            p = (torch.rand(8, 3, 8192, generator=g) * 30).to(dev)
            q = p[:, :, :4096].contiguous()
        else:
            p = torch.rand(4, 3, 4096, device=dev) * 30
            q = torch.rand(4, 3, 4096, device=dev) * 30
        step = lambda: k_nearest_neighbor(p, q, k)
    elif which == "knn2d":  # the nearest projected point of every pixel at level 1, both frames of a batch of 4 (RPEFlow_core.py:327-330)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from tests import inputs as I
        r = I.rng(0)
        p = torch.from_numpy(I.pixel_cloud(r, 8, 4096, 144, 240)).to(dev).transpose(1, 2).contiguous()
        q = torch.from_numpy(I.pixel_grid(8, 144, 240)).to(dev).transpose(1, 2).contiguous()
        step = lambda: k_nearest_neighbor(p, q, 1)
    elif which == "corr3d":  # Correlation3D at level 1: B=4, N=4096, C=32
        from rpeflow_amd.pwc3d_core import Correlation3D
        m = Correlation3D(32, 32).to(dev).eval()
        xyz1 = torch.randn(4, 3, 4096, device=dev)
        xyz2 = xyz1 + 0.05 * torch.randn_like(xyz1)
        f1, f2 = torch.randn(4, 32, 4096, device=dev), torch.randn(4, 32, 4096, device=dev)
        step = lambda: m(xyz1, f1, xyz2, f2)
    elif which == "project":  # project_feat_with_nn_corr at level 1: B=4, 144x240, N=4096, C2=81, C3=37 (RPEFlow_core.py:78-83)
        from rpeflow_amd import utils as U
        B, H, W, N, C2, C3 = 4, 144, 240, 4096, 81, 37
        feat2d, feat3d = torch.randn(B, C2, H, W, device=dev), torch.randn(B, C3, N, device=dev)
        xy = torch.stack([torch.rand(B, N, device=dev) * W, torch.rand(B, N, device=dev) * H], 1)
        nn = k_nearest_neighbor(xy, U.mesh_grid(B, H, W, dev).reshape(B, 2, -1), 1)[..., 0]
        sampled = U.grid_sample_wrapper(feat2d, xy)  # what the 3-D fuser of the pair has already (the decoder levels pass it on)
        step = lambda: (U.project_feat_with_nn_corr(xy, feat2d, feat3d, nn), U.project_feat_with_nn_corr(xy, feat2d, feat3d, nn, sampled_2d=sampled))
    elif which == "pw_l1":  # the level-1 cross block's GDFN project_in: 96 -> 510 channels over 4 x 144 x 240 (restormer_arch.py:207-222), 13.5 GFLOP
        from rpeflow_amd import utils as U
        conv = torch.nn.Conv2d(96, 510, 1, bias=False).to(dev)
        x = torch.randn(4, 96, 144, 240, device=dev)
        step = lambda: U.conv_module(conv, x)
    else:
        raise SystemExit("unknown operator " + which)
    for _ in range(iters):
        step()
torch.cuda.synchronize()
print("done", which, iters)
