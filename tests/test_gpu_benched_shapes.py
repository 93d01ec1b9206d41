"""The kernel instantiations bench.py's configuration launches (B = 4, FlyingThings3D pyramid),
each against the CPU oracle DIRECTLY -- not only through the whole model's EPE: a mean-EPE bound of 1e-4 over 8192 points can
hide a wrong tile.  Reference semantics: models/pointconv.py:90-122, models/pwc3d_core.py:69-117, models/utils.py:140-156,
297-317."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import inputs as I
from tests.test_gpu_glue import DEV, _load, close, close_sum, dev
from tests.test_oracle_golden import _shapes_corr3d, _shapes_pointconv

pytestmark = pytest.mark.gpu

from rpeflow_amd import pointconv as PC  # noqa: E402
from rpeflow_amd import pwc3d_core as P3  # noqa: E402
from rpeflow_amd import utils as U  # noqa: E402

B = 4


def cloud(r, n, batch=B):
    return np.ascontiguousarray(I.ids_cloud(r, batch, n).transpose(0, 2, 1))


@torch.no_grad()
@pytest.mark.parametrize("C,Cout,N", [(195, 128, 4096), (128, 128, 4096), (195, 128, 2048), (128, 128, 256)])
def test_flow_estimator_pointconv_layers_at_the_benched_sizes(C, Cout, N):
    """FlowEstimator3D's conv1 (195 -> 128) and conv2 (128 -> 128) at level 1 (N = 4096: the XCD-remapped 1-D grid of
    pointconv_fused_kernel<1,2,4,2,1>), level 2 and the smallest level, B = 4, norm None (pwc3d_core.py:123-124)."""
    r = I.rng(8100 + C + N)
    xyz = cloud(r, N)
    feat = r.standard_normal((B, C, N), dtype=np.float32)
    knn = O.k_nearest_neighbor(xyz, xyz, 16)
    m, p = _load(PC.PointConvNoSampling(C, Cout, norm=None), _shapes_pointconv(C, Cout, None), 8100 + C)
    out = m(dev(xyz), dev(feat), dev(knn))
    ref = O.pointconv(p, xyz, feat, knn_indices=knn, k=16, norm=None)
    close_sum(out, ref, what="PointConvNoSampling %d -> %d, N = %d" % (C, Cout, N))
    # every sample and every query tile was written by its own workgroup: no row may be a copy of another sample's
    got = out.cpu().numpy()
    assert not np.array_equal(got[0], got[1]) and np.isfinite(got).all()


@torch.no_grad()
@pytest.mark.parametrize("C,M,Q", [(32, 8192, 4096), (64, 4096, 2048)])
def test_pyramid_pointconv_down_sampling_at_the_benched_sizes(C, M, Q):
    """FeaturePyramid3D's PointConvDownSampling (pointconv.py:33-61) with eval-mode BatchNorm at the two largest levels,
    2B = 8 clouds as the forward batches the two frames."""
    r = I.rng(8200 + C)
    xyz = cloud(r, M, 2 * B)
    feat = r.standard_normal((2 * B, C, M), dtype=np.float32)
    sampled = np.ascontiguousarray(xyz[:, :, :Q])
    m, p = _load(PC.PointConvDownSampling(C, C, norm="batch_norm"), _shapes_pointconv(C, C, "batch_norm"), 8200 + C)
    out = m(dev(xyz), dev(feat), dev(sampled))
    close_sum(out, O.pointconv(p, xyz, feat, sampled_xyz=sampled, k=16, norm="batch_norm"))


@torch.no_grad()
@pytest.mark.parametrize("C,N", [(32, 4096), (64, 2048), (192, 256)])
def test_correlation3d_at_the_benched_sizes(C, N):
    """corr3d_cost_kernel / corr3d_n2n_kernel at (N, C) = (4096, 32), (2048, 64), (256, 192), B = 4."""
    r = I.rng(8300 + C)
    xyz1 = cloud(r, N)
    xyz2 = (xyz1 + r.standard_normal(xyz1.shape, dtype=np.float32) * np.float32(0.2)).astype(np.float32)
    feat1 = r.standard_normal((B, C, N), dtype=np.float32)
    feat2 = r.standard_normal((B, C, N), dtype=np.float32)
    m, p = _load(P3.Correlation3D(C, C, k=16), _shapes_corr3d(C), 8300 + C)
    knn11 = O.k_nearest_neighbor(xyz1, xyz1, 16)
    out = m(dev(xyz1), dev(feat1), dev(xyz2), dev(feat2), dev(knn11))
    close_sum(out, O.correlation3d(p, xyz1, feat1, xyz2, feat2, knn11, 16))


def test_project_feat_with_nn_corr_at_level_1():
    """project_rows_kernel / point_rows_kernel at 144 x 240, N = 4096, C2 = 81 (the correlation fuser's cost volume),
    C3 = 37, B = 4; nearest points from the oracle's own search and from the operator's implicit one."""
    r = I.rng(8400)
    H, W, N, C2, C3 = 144, 240, 4096, 81, 37
    feat2 = I.feature_map(r, B, C2, H, W)
    xy = np.ascontiguousarray(I.pixel_cloud(r, B, N, H, W).transpose(0, 2, 1))
    feat3 = r.standard_normal((B, C3, N), dtype=np.float32)
    nn = O.k_nearest_neighbor(xy, np.ascontiguousarray(I.pixel_grid(B, H, W).transpose(0, 2, 1)), 1)[..., 0]
    ref = O.project_feat_with_nn_corr(xy, feat2, feat3, nn)
    got = U.project_feat_with_nn_corr(dev(xy), dev(feat2), dev(feat3), dev(nn))
    close(got, ref, 3e-6, what="explicit nearest-point table")
    assert torch.equal(got, U.project_feat_with_nn_corr(dev(xy), dev(feat2), dev(feat3)))  # the implicit search picks the same points
    sampled = U.grid_sample_wrapper(dev(feat2), dev(xy))
    close(sampled, O.grid_sample_wrapper(feat2, xy), 3e-6, what="grid_sample_wrapper C = 81, N = 4096")
    assert torch.equal(got, U.project_feat_with_nn_corr(dev(xy), dev(feat2), dev(feat3), dev(nn), sampled_2d=sampled))


def test_backwarp_2d_at_level_1():
    """bilinear_kernel on [4, 32, 144, 240] (RPEFlow_core.py:349: the largest warp of the forward)."""
    r = I.rng(8450)
    feat = I.feature_map(r, B, 32, 144, 240)
    flow = I.flow_field(r, B, 144, 240, std=6.0)
    close(U.backwarp_2d(dev(feat), dev(flow), "border"), O.backwarp_2d(feat, flow, "border"), 3e-6)


def test_knn_interpolation_at_the_top_level():
    """knn_interp_kernel 4096 -> 8192 queries, C = 67 (RPEFlow_core.py:355 at level 0) and C = 3 (the flow, :430), B = 4."""
    r = I.rng(8500)
    xyz, q = cloud(r, 4096), cloud(r, 8192)
    for C in (67, 3):
        f = r.standard_normal((B, C, 4096), dtype=np.float32)
        close(U.knn_interpolation(dev(xyz), dev(f), dev(q), k=3), O.knn_interpolation(xyz, f, q, 3), 1e-5, what="C = %d" % C)
