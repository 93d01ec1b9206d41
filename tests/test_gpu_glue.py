"""GPU parity tests for the glue ops of models/utils.py and the 3-D blocks of
models/pointconv.py / models/pwc3d_core.py, through the C ABI."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import cases as K
from tests import inputs as I
from tests.test_oracle_golden import _shapes_corr3d, _shapes_pointconv

pytestmark = pytest.mark.gpu

from rpeflow_amd import utils as U  # noqa: E402
from rpeflow_amd import pointconv as PC  # noqa: E402
from rpeflow_amd import pwc3d_core as P3  # noqa: E402

DEV = "cuda:0"


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(got, ref, atol, rtol=0.0, what=""):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=what)


def close_sum(got, ref, what=""):
    """For outputs that are fp32 sums of thousands of terms (PointConv linear over 16*(C+3)
    inputs, Correlation3D): error scales with the magnitude of the summands, so the absolute
    tolerance is 2e-6 of the largest output plus 1e-4 relative."""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-6 * float(np.abs(ref).max()) + 1e-6, err_msg=what)


def test_glue_ops_against_reference_golden(golden_dir):
    d, g = K.glue_inputs(), G(golden_dir, "glue_ops")
    t = {k: dev(v) for k, v in d.items()}
    assert np.array_equal(U.batch_indexing_channel_first(t["feat_3d"], t["idx"]).cpu().numpy(), g["gather_cf"])
    cl = t["feat_3d"].transpose(1, 2).contiguous()
    assert np.array_equal(U.batch_indexing_channel_last(cl, t["idx"]).cpu().numpy(), g["gather_cl"])
    # strided (non-contiguous) source, as the model's transposed views are
    assert np.array_equal(U.batch_indexing_channel_last(t["feat_3d"].transpose(1, 2), t["idx"]).cpu().numpy(), g["gather_cl"])
    close(U.backwarp_2d(t["feat_2d"], t["flow"], padding_mode="border"), g["backwarp_2d"], 3e-6, what="backwarp_2d")
    close(U.grid_sample_wrapper(t["feat_2d"], t["xy"]), g["grid_sample_wrapper"], 3e-6, what="grid_sample_wrapper")
    close(U.knn_interpolation(t["xyz"], t["feat_3d"], t["xyz_q"], k=3), g["knn_interp"], 5e-6, what="knn_interpolation")
    close(U.backwarp_3d(t["xyz"], t["xyz"] + 0.1, t["flow3"], k=3), g["backwarp_3d"], 1e-5, what="backwarp_3d")
    close(U.project_feat_with_nn_corr(t["xy"], t["feat_2d"], t["feat_3d"]), g["project_feat"], 3e-6, what="project_feat")


def test_glue_ops_against_oracle_other_shapes():
    r = I.rng(5100)
    B, C2, C3, H, W, N = 3, 33, 67, 9, 15, 256
    feat2 = I.feature_map(r, B, C2, H, W)
    flow = I.flow_field(r, B, H, W, std=6.0)
    xy = np.ascontiguousarray(I.pixel_cloud(r, B, N, H, W).transpose(0, 2, 1))
    feat3 = r.standard_normal((B, C3, N), dtype=np.float32)
    close(U.backwarp_2d(dev(feat2), dev(flow), "border"), O.backwarp_2d(feat2, flow, "border"), 3e-6)
    close(U.backwarp_2d(dev(feat2), dev(flow), "zeros"), O.backwarp_2d(feat2, flow, "zeros"), 3e-6)
    close(U.grid_sample_wrapper(dev(feat2), dev(xy)), O.grid_sample_wrapper(feat2, xy), 3e-6)
    nn = O.k_nearest_neighbor(xy, np.ascontiguousarray(I.pixel_grid(B, H, W).transpose(0, 2, 1)), 1)[..., 0]
    got = U.project_feat_with_nn_corr(dev(xy), dev(feat2), dev(feat3), dev(nn))
    close(got, O.project_feat_with_nn_corr(xy, feat2, feat3, nn), 3e-6)
    # the implicit KNN (nn_indices=None) picks the same pixels
    got2 = U.project_feat_with_nn_corr(dev(xy), dev(feat2), dev(feat3))
    assert torch.equal(got, got2)
    # interpolation with C=67 features and 8192 queries (RPEFlow_core.py:355 at level 0)
    xyz = np.ascontiguousarray(I.ids_cloud(r, 2, 4096).transpose(0, 2, 1))
    q = np.ascontiguousarray(I.ids_cloud(r, 2, 8192).transpose(0, 2, 1))
    f = r.standard_normal((2, 67, 4096), dtype=np.float32)
    close(U.knn_interpolation(dev(xyz), dev(f), dev(q), k=3), O.knn_interpolation(xyz, f, q, 3), 1e-5)


def test_gather_edge_cases():
    r = I.rng(5200)
    data = r.standard_normal((2, 5, 11), dtype=np.float32)
    idx = r.integers(0, 11, (2, 4096, 16)).astype(np.int64)
    assert np.array_equal(U.batch_indexing_channel_first(dev(data), dev(idx)).cpu().numpy(), O.batch_indexing_channel_first(data, idx))
    one = r.integers(0, 11, (2, 1)).astype(np.int64)
    assert np.array_equal(U.batch_indexing_channel_first(dev(data), dev(one)).cpu().numpy(), O.batch_indexing_channel_first(data, one))
    flat = r.standard_normal((2, 11), dtype=np.float32)  # 2-D data branch, utils.py:113-114
    assert np.array_equal(U.batch_indexing_channel_last(dev(flat), dev(one)).cpu().numpy(), O.batch_indexing_channel_last(flat, one))
    # prefix slice of indices (sample_index[:, :n], pwc3d_core.py:25)
    wide = r.integers(0, 11, (2, 40)).astype(np.int64)
    got = U.batch_indexing_channel_first(dev(data), dev(wide)[:, :17]).cpu().numpy()
    assert np.array_equal(got, O.batch_indexing_channel_first(data, wide[:, :17]))


def _load(module, shapes, seed):
    params = I.fill_params(shapes, seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return module.to(DEV).eval(), params


@torch.no_grad()
@pytest.mark.parametrize("name", ["pointconv_down", "pointconv_nosample"])
def test_pointconv_modules(golden_dir, name):
    c, x = K.BLOCK_CASES[name], K.block_inputs(name)
    cls = PC.PointConvDownSampling if name == "pointconv_down" else PC.PointConvNoSampling
    m, p = _load(cls(c["C"], c["Cout"], norm=c["norm"], k=c["k"]), _shapes_pointconv(c["C"], c["Cout"], c["norm"]), c["seed"] + 1000)
    if name == "pointconv_down":
        out = m(dev(x["xyz"]), dev(x["feat"]), dev(x["sampled"]))
    else:
        out = m(dev(x["xyz"]), dev(x["feat"]))
        knn = O.k_nearest_neighbor(x["xyz"], x["xyz"], 20)  # wider precomputed table, pointconv.py:102-105
        out2 = m(dev(x["xyz"]), dev(x["feat"]), dev(knn))
        assert torch.equal(out, out2)
    close_sum(out, G(golden_dir, name)["out"], what=name + " vs reference golden")
    ref = O.pointconv(p, x["xyz"], x["feat"], sampled_xyz=x["sampled"], k=c["k"], norm=c["norm"])
    close_sum(out, ref, what=name + " vs oracle")


@torch.no_grad()
@pytest.mark.parametrize("C,M,Q", [(2, 64, 64), (61, 500, 77), (125, 300, 300), (195, 256, 256), (253, 100, 40)])
def test_pointconv_channel_counts(C, M, Q):
    """Channel counts around the 16-channel chunk boundaries (C + 3 = 5 ... 256) against the oracle."""
    r = I.rng(6100 + C)
    xyz = np.ascontiguousarray(I.ids_cloud(r, 2, M).transpose(0, 2, 1))
    feat = r.standard_normal((2, C, M), dtype=np.float32)
    m, p = _load(PC.PointConvDownSampling(C, 8, norm=None), _shapes_pointconv(C, 8, None), 77)
    sampled = xyz[:, :, :Q].copy()
    out = m(dev(xyz), dev(feat), dev(sampled))
    close_sum(out, O.pointconv(p, xyz, feat, sampled_xyz=sampled, k=16))


@pytest.mark.parametrize("name", list(K.GENERAL_CASES))
def test_pointconv_and_correlation3d_outside_the_fused_configuration(golden_dir, name):
    """Everything the reference's constructors accept (pointconv.py:8-31, pwc3d_core.py:61-67) that the fused kernels are not
    built for -- k = 5 / 9 / 20, instance_norm, relu / no activation, training-mode BatchNorm (with gradients), a 200-channel
    Correlation3D -- runs the reference's op sequence on the GPU (HIP neighbour search and gathers): against the reference's
    own outputs (tests/golden/general_modules.npz, made by importing it)."""
    c, x = K.GENERAL_CASES[name], K.block_inputs(name)
    g = G(golden_dir, "general_modules")
    if c["kind"] == "corr":
        m = P3.Correlation3D(c["C"], c["Cout"], k=c["k"])
        assert not m.fusable
        shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        m, _ = _load(m, shapes, c["seed"] + 1000)
        with torch.no_grad():
            out = m(dev(x["xyz1"]), dev(x["feat1"]), dev(x["xyz2"]), dev(x["feat2"]))
        close_sum(out, g[name], what=name)
        return
    cls = PC.PointConvDownSampling if c["kind"] == "down" else PC.PointConvNoSampling
    m = cls(c["C"], c["Cout"], norm=c["norm"], activation=c["activation"], k=c["k"])
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    m, _ = _load(m, shapes, c["seed"] + 1000)
    m.train(c["train"])
    args = [dev(x["xyz"]), dev(x["feat"])] + ([dev(x["sampled"])] if c["kind"] == "down" else [])
    if not c["train"]:
        assert not m.fusable
        with torch.no_grad():
            close_sum(m(*args), g[name], what=name)
            rows = m(*args, out_rows=True)  # (the next layer's gather rows, as FlowEstimator3D asks for them)
            close_sum(rows.rows[:, :, 3:3 + c["Cout"]].transpose(1, 2), g[name], what=name + " rows")
        return
    # training mode: batch statistics, running-stat update, gradients through the gathers (torch.gather there) and the GEMMs
    args[1].requires_grad_(True)
    y = m(*args)
    (y * y).sum().backward()
    close_sum(y.detach(), g[name], what=name)
    for got, want, what in ((args[1].grad, g[name + "__grad_feat"], "d/d features"), (m.linear.weight.grad, g[name + "__grad_linear"], "d/d linear.weight")):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-3, atol=2e-5 * float(np.abs(want).max()), err_msg=name + " " + what)
    close_sum(m.norm_fn.running_mean, g[name + "__running_mean"], what=name + " running mean")
    m.eval()  # ... and back under no_grad with the norm in eval mode the fused kernel runs again
    with torch.no_grad():
        fused = m(*args)
        from rpeflow_amd.csrc import k_nearest_neighbor
        general = m._general(args[0], args[1].detach(), args[0], k_nearest_neighbor(args[0], args[0], c["k"]), False)
    close_sum(fused, general.cpu().numpy(), what=name + " fused vs general in eval mode")


@torch.no_grad()
def test_correlation3d_module(golden_dir):
    c, x = K.BLOCK_CASES["correlation3d"], K.block_inputs("correlation3d")
    m, p = _load(P3.Correlation3D(c["C"], c["C"], k=c["k"]), _shapes_corr3d(c["C"]), c["seed"] + 1000)
    out = m(dev(x["xyz1"]), dev(x["feat1"]), dev(x["xyz2"]), dev(x["feat2"]))
    close_sum(out, G(golden_dir, "correlation3d")["out"], what="vs reference golden")
    knn11 = O.k_nearest_neighbor(x["xyz1"], x["xyz1"], c["k"])
    out2 = m(dev(x["xyz1"]), dev(x["feat1"]), dev(x["xyz2"]), dev(x["feat2"]), dev(knn11))
    close_sum(out2, O.correlation3d(p, x["xyz1"], x["feat1"], x["xyz2"], x["feat2"], knn11, c["k"]))


@torch.no_grad()
@pytest.mark.parametrize("C,N", [(32, 150), (64, 130), (96, 70), (128, 97), (192, 61)])
def test_correlation3d_every_pyramid_width(C, N):
    """The model's Correlation3D widths (RPEFlow_core.py:228-234): every instantiation of the cost / n2n kernels -- C = 192 runs
    the cost kernel with the four waves of a workgroup sharing one point -- against the oracle."""
    r = I.rng(9100 + C)
    xyz1 = np.ascontiguousarray(I.ids_cloud(r, 2, N).transpose(0, 2, 1))
    xyz2 = (xyz1 + r.standard_normal(xyz1.shape, dtype=np.float32) * np.float32(0.2)).astype(np.float32)
    feat1 = r.standard_normal((2, C, N), dtype=np.float32)
    feat2 = r.standard_normal((2, C, N), dtype=np.float32)
    m, p = _load(P3.Correlation3D(C, C, k=16), _shapes_corr3d(C), 4000 + C)
    out = m(dev(xyz1), dev(feat1), dev(xyz2), dev(feat2))
    close_sum(out, O.correlation3d(p, xyz1, feat1, xyz2, feat2, k=16))


def _own_shapes(module):
    """(key, shape) in state-dict order: the reference's order when the module tree mirrors the reference's."""
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


@torch.no_grad()
def test_flow_estimator3d_module(golden_dir):
    """pwc3d_core.py:120-148 (two PointConvNoSampling on a shared neighbour table, MLP, 1x1 head) against the reference."""
    c, x = K.BLOCK_CASES["flow_estimator3d"], K.block_inputs("flow_estimator3d")
    m = P3.FlowEstimator3D(c["channels"], k=c["k"])
    m, _ = _load(m, _own_shapes(m), c["seed"] + 1000)
    xyz = dev(x["xyz"])
    from rpeflow_amd.csrc import k_nearest_neighbor
    knn = k_nearest_neighbor(xyz, xyz, k=c["k"])
    feat, flow = m(xyz, dev(x["feat"]), knn)
    g = G(golden_dir, "flow_estimator3d")
    close_sum(feat, g["feat"], what="features vs reference golden")
    close_sum(flow, g["flow"], what="flow vs reference golden")


@torch.no_grad()
def test_feature_pyramid3d_module(golden_dir):
    """build_pc_pyramid + FeaturePyramid3D (pwc3d_core.py:8-57): sample indices exactly, every level's features."""
    c, x = K.BLOCK_CASES["feature_pyramid3d"], K.block_inputs("feature_pyramid3d")
    m = P3.FeaturePyramid3D(c["channels"], norm=c["norm"], k=c["k"])
    m, _ = _load(m, _own_shapes(m), c["seed"] + 1000)
    xyzs1, xyzs2, idx1, idx2 = P3.build_pc_pyramid(dev(x["pc1"]), dev(x["pc2"]), c["samples"])
    g = G(golden_dir, "feature_pyramid3d")
    for i in range(len(c["samples"]) + 1):
        assert np.array_equal(idx1[i].cpu().numpy(), g["index1_%d" % i]) and np.array_equal(idx2[i].cpu().numpy(), g["index2_%d" % i])
    feats = m(xyzs1)
    assert len(feats) == len(c["channels"])
    for i, f in enumerate(feats):
        close_sum(f, g["feat%d" % i], what="level %d vs reference golden" % i)


@torch.no_grad()
def test_build_pc_pyramid_prefix_property():
    r = I.rng(6200)
    pc1 = np.ascontiguousarray(I.ids_cloud(r, 2, 8192).transpose(0, 2, 1))
    pc2 = (pc1 + r.standard_normal(pc1.shape, dtype=np.float32) * np.float32(0.05)).astype(np.float32)
    xyzs1, xyzs2, idx1, idx2 = P3.build_pc_pyramid(dev(pc1), dev(pc2), [4096, 2048, 1024, 512, 256])
    both = np.concatenate([pc1, pc2]).transpose(0, 2, 1)
    ref = O.furthest_point_sampling(both, 4096)
    assert np.array_equal(idx1[1].cpu().numpy(), ref[:2]) and np.array_equal(idx2[1].cpu().numpy(), ref[2:])
    for lvl, n in enumerate([4096, 2048, 1024, 512, 256], start=1):
        assert xyzs1[lvl].shape == (2, 3, n)
        assert np.array_equal(xyzs1[lvl].cpu().numpy(), np.take_along_axis(pc1, ref[:2, None, :n], axis=2))
        assert np.array_equal(xyzs2[lvl].cpu().numpy(), np.take_along_axis(pc2, ref[2:, None, :n], axis=2))


@pytest.mark.parametrize("name", list(K.EVENT_CASES))
def test_events_to_voxel_matches_reference_bit_for_bit(golden_dir, name):
    """On-device voxelisation (event_utils.eventsToVoxel): per-pixel sums in event order, so the
    fp32 result equals the reference's CPU index_put_ accumulation exactly."""
    from rpeflow_amd.event_ops import events_to_voxel
    ev, H, W, bins, pol = K.event_inputs(name)
    got = events_to_voxel(torch.from_numpy(ev).to("cuda:0"), num_bins=bins, height=H, width=W, event_polarity=pol).cpu().numpy()
    ref = np.load(os.path.join(golden_dir, name + ".npz"))["voxel"]
    assert got.shape == ref.shape and np.array_equal(np.isnan(got), np.isnan(ref))  # (the same-timestamp cases: NaN where the reference has NaN)
    assert np.array_equal(np.nan_to_num(got).view(np.uint32), np.nan_to_num(ref).view(np.uint32))
    again = events_to_voxel(torch.from_numpy(ev).to("cuda:0"), num_bins=bins, height=H, width=W, event_polarity=pol).cpu().numpy()
    assert np.array_equal(got, again, equal_nan=True)  # no atomics: repeatable


def test_events_to_voxel_edge_cases():
    from rpeflow_amd.event_ops import events_to_voxel
    empty = events_to_voxel(torch.zeros(0, 4, dtype=torch.float64, device="cuda:0"), num_bins=3, height=4, width=5, event_polarity=True)
    assert empty.shape == (6, 4, 5) and float(empty.abs().sum()) == 0.0
    r = I.rng(77)
    ev = np.stack([r.integers(0, 7, 50), r.integers(0, 3, 50), np.sort(r.random(50)) * 1e4, r.integers(-1, 2, 50)], 1).astype(np.float64)
    got = events_to_voxel(torch.from_numpy(ev).to("cuda:0"), num_bins=4, height=3, width=7, event_polarity=False).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), O.events_to_voxel(ev, 4, 3, 7, False).view(np.uint32))  # signed polarity weights
    with pytest.raises(IndexError):
        events_to_voxel(torch.tensor([[9.0, 0.0, 0.0, 1.0], [0.0, 0.0, 1.0, 1.0]], dtype=torch.float64, device="cuda:0"), 2, 3, 7)


def test_events_to_voxel_full_size_conservation():
    """One million events on a 540x960 sensor: every event spreads a total weight of exactly 1 over two neighbouring time
    bins of its polarity grid, so grid sums count the events (up to fp32 summation error), and each event lands in its pixel."""
    from rpeflow_amd.event_ops import events_to_voxel
    r = I.rng(91)
    n, H, W, bins = 1_000_000, 540, 960, 10
    ev = np.stack([r.integers(0, W, n), r.integers(0, H, n), np.sort(r.integers(0, 50_000, n)), r.integers(0, 2, n)], 1).astype(np.float64)
    vox = events_to_voxel(torch.from_numpy(ev).to("cuda:0"), num_bins=bins, height=H, width=W, event_polarity=True)
    assert vox.shape == (2 * bins, H, W) and bool((vox >= 0).all())
    pos, neg = float(vox[:bins].double().sum()), float(vox[bins:].double().sum())
    n_pos = int((ev[:, 3] > 0).sum())
    assert abs(pos - n_pos) < 1e-3 * n and abs(neg - (n - n_pos)) < 1e-3 * n
    per_pixel = torch.zeros(H * W, dtype=torch.float64).index_add_(0, torch.from_numpy((ev[:, 1] * W + ev[:, 0]).astype(np.int64)), torch.ones(n, dtype=torch.float64))
    assert float((vox.double().sum(0).cpu().reshape(-1) - per_pixel).abs().max()) < 1e-3


@pytest.mark.parametrize("B,Ca,Cb,h,w", [(4, 2, 32, 9, 15), (2, 2, 32, 18, 30), (1, 3, 5, 1, 7), (2, 2, 0, 36, 60), (1, 1, 1, 1, 1)])
def test_upsample2x_pair_matches_interpolate(B, Ca, Cb, h, w):
    """The decoder's coarse-to-fine hand-over in one launch == F.interpolate(a*2, x2), F.interpolate(b, x2) (align_corners)."""
    import torch.nn.functional as F
    from rpeflow_amd.utils import upsample2x_pair
    g = torch.Generator().manual_seed(B * 100 + h)
    a, b = torch.randn(B, Ca, h, w, generator=g), torch.randn(B, Cb, h, w, generator=g)
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)
    want_a, want_b = up(a * 2), (up(b) if Cb else b.new_zeros(B, 0, 2 * h, 2 * w))   # CPU: the reference's path
    got_a, got_b = upsample2x_pair(a.cuda(), b.cuda(), scale_a=2.0)
    assert got_a.shape == want_a.shape and got_b.shape == want_b.shape
    assert (got_a.cpu() - want_a).abs().max().item() <= 2e-6 * max(1.0, want_a.abs().max().item())
    if Cb:
        assert (got_b.cpu() - want_b).abs().max().item() <= 2e-6 * max(1.0, want_b.abs().max().item())


@pytest.mark.parametrize("B,H,W", [(2, 40, 100), (1, 64, 128), (3, 30, 64)])
def test_resize_frames_matches_the_reference_preparation(B, H, W):
    """RPEFlow.forward's input preparation (RPEFlow.py:40-47): images.float()/255 and event grid resized to multiples of 64
    (utils.py:227-241), frames split -- one launch each, against the same steps on the CPU."""
    import torch.nn.functional as F
    from rpeflow_amd.utils import resize_frames
    g = torch.Generator().manual_seed(H)
    images = torch.randint(0, 256, (B, 6, H, W), generator=g, dtype=torch.uint8)
    events = torch.randn(B, 5, H, W, generator=g)
    size = ((H + 63) // 64 * 64, (W + 63) // 64 * 64)
    up = lambda t: t if tuple(t.shape[2:]) == size else F.interpolate(t, size=size, mode="bilinear", align_corners=True)
    ref = up(images.float() / 255.0)
    want_both, want_events = torch.cat([ref[:, :3], ref[:, 3:]], dim=0), up(events)
    got_both = resize_frames(images.cuda(), size, divisor=255.0, pair_split=True).cpu()
    got_float = resize_frames(images.float().cuda(), size, divisor=255.0, pair_split=True).cpu()
    got_events = resize_frames(events.cuda(), size).cpu()
    assert got_both.shape == want_both.shape and torch.equal(got_both, got_float)
    assert (got_both - want_both).abs().max().item() <= 1e-6 and (got_events - want_events).abs().max().item() <= 2e-6 * events.abs().max().item()
    if size == (H, W):
        assert torch.equal(got_both, want_both) and torch.equal(got_events, want_events)


# ---------------------------------------------------------------- IDS transforms and the final flow resize
@pytest.mark.parametrize("H,W,seeds,dsec", [(544, 960, [1000, 1001, 1002], False), (480, 640, [2000], True), (128, 192, [7], False)])
def test_ids_forward_bit_exact_against_oracle(H, W, seeds, dsec):
    """rpe_ids_forward (csrc/ids.hip) against the CPU restatement of perspect2parallel: every coordinate bit for bit --
    furthest-point sampling behind it is chaotic in these values."""
    from rpeflow_amd.model import RPEFlow
    samples = [I.frame_pair(s, H=H, W=W, N=8192, dsec=dsec) for s in seeds]
    pcs, intr = np.stack([s["pcs"] for s in samples]), np.stack([s["intrinsics"] for s in samples])
    model = RPEFlow()
    inputs = {"images": torch.zeros(len(seeds), 6, H, W, dtype=torch.uint8), "pcs": dev(pcs), "intrinsics": dev(intr)}
    persp, paral = model._cameras(inputs)
    got = U.ids_forward(inputs["pcs"], inputs["intrinsics"], persp, paral).cpu().numpy()
    Hp, Wp = paral["sensor_h"], paral["sensor_w"]
    want = np.concatenate([O.perspect2parallel(pcs[:, :3], intr, H, W, Hp, Wp), O.perspect2parallel(pcs[:, 3:], intr, H, W, Hp, Wp)])
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # strided input (a channel-sliced view) gives the same
    wide = torch.zeros(len(seeds), 8, 8192, device=DEV)
    wide[:, 1:7] = inputs["pcs"]
    assert torch.equal(U.ids_forward(wide[:, 1:7], inputs["intrinsics"], persp, paral), torch.from_numpy(got).to(DEV))


def test_ids_forward_against_the_reference_golden(golden_dir):
    """... and against the clouds the reference itself produced for the benched batch (its CPU log is not correctly rounded
    on a few values in ten thousand: at most one ulp on those, everything else identical)."""
    from rpeflow_amd.model import RPEFlow
    g = G(golden_dir, "model_bench_b4_544x960")
    samples = [I.frame_pair(1000 + i, H=544, W=960, N=8192) for i in range(4)]
    inputs = {"images": torch.zeros(4, 6, 544, 960, dtype=torch.uint8), "pcs": dev(np.stack([s["pcs"] for s in samples])),
              "intrinsics": dev(np.stack([s["intrinsics"] for s in samples]))}
    model = RPEFlow()
    got = U.ids_forward(inputs["pcs"], inputs["intrinsics"], *model._cameras(inputs)).cpu().numpy()
    want = np.concatenate([g["pc1_ids"], g["pc2_ids"]])
    ulps = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert np.array_equal(got[:, :2].view(np.uint32), want[:, :2].view(np.uint32))
    assert ulps.max() <= 1 and (ulps != 0).sum() <= 8, (ulps.max(), (ulps != 0).sum())


def test_default_ids_and_sampling_against_64_reference_clouds(golden_dir):
    """The product's DEFAULT front end -- rpe_ids_forward + rpe_fps on the device -- against what it replaces in the reference
    (host-side perspect2parallel, utils.py:320-346 via RPEFlow.py:56-69, then build_pc_pyramid's furthest-point sampling,
    pwc3d_core.py:11-13), on 64 clouds of 32 large-motion frame pairs (tests/golden/ids_fps_sweep.npz: the reference's full
    sampling orders, and its z' wherever torch.log on the build container's CPU was not the correctly rounded logarithm).
    x', y' bit for bit; z' = (f log z + 1) s differs from the reference exactly where the reference's log is off the correctly
    rounded value (one ulp of log z, which the product with f and the two roundings behind it turn into at most two ulps of
    z'); and the sampling order -- chaotic in these values -- is counted cloud by cloud.  Measured: 32 of 524 288 z' apart
    (28 clouds touched), 0 of 64 sampling orders different."""
    from rpeflow_amd.csrc import furthest_point_sampling
    from rpeflow_amd.model import RPEFlow
    g = G(golden_dir, "ids_fps_sweep")
    H, W, N = 544, 960, 8192
    model = RPEFlow()
    off_values, touched, divergent = 0, 0, []
    for first in range(0, 32, 4):  # batches of four frame pairs = eight clouds, as the forward sees them
        samples = [I.frame_pair_stress(6000 + first + i, H=H, W=W, N=N) for i in range(4)]
        inputs = {"images": torch.zeros(4, 6, H, W, dtype=torch.uint8), "pcs": dev(np.stack([s["pcs"] for s in samples])),
                  "intrinsics": dev(np.stack([s["intrinsics"] for s in samples]))}
        clouds = U.ids_forward(inputs["pcs"], inputs["intrinsics"], *model._cameras(inputs))  # [8,3,N]: frame-1 clouds, then frame-2
        order = furthest_point_sampling(clouds.transpose(1, 2), 4096).cpu().numpy()
        assert np.array_equal(order, model.sample_order(inputs).cpu().numpy())  # what forward() / forward_ahead() consume
        got = clouds.cpu().numpy()
        for j in range(8):
            c = 2 * (first + j % 4) + j // 4  # the sweep's cloud number: pair-major, frame-minor
            pcs, intr = samples[j % 4]["pcs"], samples[j % 4]["intrinsics"]
            ref = O.perspect2parallel(pcs[None, 3 * (j // 4):3 * (j // 4) + 3], intr[None], H, W, 18, 30)[0]
            assert np.array_equal(got[j].view(np.uint32), ref.view(np.uint32))  # the device computes the correctly rounded form
            m = g["patch_cloud"] == c
            ref[2][g["patch_pos"][m]] = g["patch_val"][m]                       # ... and this is the cloud the reference computed
            ulps = np.abs(got[j].view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
            assert ulps.max() <= 2 and int((ulps != 0).sum()) == int(m.sum()) and not ulps[:2].any()
            off_values += int(m.sum())
            touched += bool(m.any())
            if not np.array_equal(order[j].astype(np.uint16), g["order"][c]):
                divergent.append((c, int(np.nonzero(order[j] != g["order"][c])[0][0])))
        # the reference's own clouds (patched values in) through the device sampling: the reference's order, all eight
        patched = got.copy()
        for j in range(8):
            m = g["patch_cloud"] == 2 * (first + j % 4) + j // 4
            patched[j, 2, g["patch_pos"][m]] = g["patch_val"][m]
        again = furthest_point_sampling(dev(patched).transpose(1, 2), 4096).cpu().numpy()
        assert np.array_equal(again.astype(np.uint16), g["order"][[2 * (first + j % 4) + j // 4 for j in range(8)]])
    print("z' values one ulp off the reference:", off_values, "in", touched, "clouds; divergent sampling orders (cloud, first position):", divergent)
    assert off_values == len(g["patch_pos"]) == 32 and touched == 28
    assert divergent == []  # the decision recorded in DESIGN.md section 2: the device path IS the default because none diverge


def test_ids_flow_inverse_against_oracle():
    from rpeflow_amd.model import RPEFlow
    H, W = 544, 960
    samples = [I.frame_pair(1000 + i, H=H, W=W, N=8192) for i in range(2)]
    pcs, intr = np.stack([s["pcs"] for s in samples]), np.stack([s["intrinsics"] for s in samples])
    xyz = O.perspect2parallel(pcs[:, :3], intr, H, W, 18, 30)
    flow = (I.rng(5).standard_normal(xyz.shape) * np.array([0.3, 0.3, 1.5])[None, :, None]).astype(np.float32)
    inputs = {"images": torch.zeros(2, 6, H, W, dtype=torch.uint8), "intrinsics": dev(intr)}
    persp, paral = RPEFlow()._cameras(inputs)
    got = U.ids_flow_inverse(dev(xyz), dev(flow), inputs["intrinsics"], persp, paral).cpu().numpy()
    want = O.parallel2perspect(xyz + flow, intr, H, W, 18, 30) - O.parallel2perspect(xyz, intr, H, W, 18, 30)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))  # same operations, correctly rounded exp on both sides


@pytest.mark.parametrize("name", ["resize_flow2d", "resize_flow2d_same"])
def test_resize_flow2d_against_reference_golden(golden_dir, name):
    """utils.py:217-224 in one launch against the reference's output (and the identity when the size already matches)."""
    c, x = K.FBLOCK_CASES[name], K.fblock_inputs(name)
    flow = dev(x["flow"])
    got = U.resize_flow2d(flow, c["th"], c["tw"])
    if name.endswith("same"):
        assert got is flow
    close(got, G(golden_dir, name)["out"], atol=3e-6, what=name)


# ---------------------------------------------------------------- round-2 fused kernels, unit level
@torch.no_grad()
@pytest.mark.parametrize("C0,C1,C2,N,norm,act", [(3, 16, 16, 100, "batch_norm", "leaky_relu"), (16, 16, 32, 777, "batch_norm", "leaky_relu"),
                                                 (13, 9, 21, 50, None, "relu"), (128, 128, 64, 1030, None, "leaky_relu"),
                                                 (96, 96, 128, 513, "batch_norm", None), (24, 24, 16, 16, None, "leaky_relu"),
                                                 # the remaining tile pairs of the split kernel (four waves over the output tiles,
                                                 # tile counts that do not divide by four, a padded second layer)
                                                 (64, 64, 96, 300, "batch_norm", "leaky_relu"), (128, 128, 192, 200, None, "relu"),
                                                 (195, 128, 128, 130, None, "leaky_relu"), (32, 40, 70, 17, "batch_norm", "leaky_relu")])
def test_fused_mlp1d_two_layers(C0, C1, C2, N, norm, act):
    """MLP1d (two Conv1dNormRelu, utils.py:65-98) in one launch against the same module on the CPU: odd channel counts,
    N not a multiple of 16, channel-first and PointConv-rows output."""
    torch.manual_seed(C0 + N)
    m = U.MLP1d(C0, [C1, C2], norm=norm, activation=act).eval()
    for blk in m.convs:
        if norm:
            blk.norm_fn.running_mean.normal_(); blk.norm_fn.running_var.uniform_(0.5, 2.0)
            blk.norm_fn.weight.data.uniform_(0.5, 1.5); blk.norm_fn.bias.data.normal_()
    x = torch.randn(2, C0, N)
    ref = m(x)
    mg = m.to(DEV)
    assert U._mlp_pack(list(mg.convs)) is not None, "shape expected to take the fused kernel"
    got = mg(x.to(DEV)).cpu()
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    # a strided (channel-sliced) input
    wide = torch.randn(2, C0 + 5, N, device=DEV)
    torch.testing.assert_close(mg(wide[:, 2:2 + C0]).cpu(), m.cpu()(wide[:, 2:2 + C0].cpu()), rtol=1e-4, atol=1e-4)
    mg = m.to(DEV)
    xyz = torch.randn(2, 3, N, device=DEV)
    rows = mg(x.to(DEV), rows_xyz=xyz)
    assert rows.channels == C2 and rows.rows.shape == (2, N, (C2 + 3 + 15) // 16 * 16)
    r = rows.rows.cpu()
    assert torch.equal(r[:, :, :3], xyz.cpu().transpose(1, 2))
    torch.testing.assert_close(r[:, :, 3:3 + C2], ref.transpose(1, 2), rtol=1e-4, atol=1e-4)
    assert (r[:, :, 3 + C2:] == 0).all()


@torch.no_grad()
@pytest.mark.parametrize("C0,C1,N", [(64, 64, 300), (5, 17, 33), (128, 64, 4096), (192, 64, 257), (60, 96, 100), (85, 128, 31), (80, 192, 50)])
def test_fused_conv1d_single_layer(C0, C1, N):
    """Conv1dNormRelu (1x1) in one launch; a layer too wide for the kernel takes the library path and still agrees."""
    torch.manual_seed(C0 * 3 + N)
    m = U.Conv1dNormRelu(C0, C1, norm="batch_norm").eval()
    m.norm_fn.running_mean.normal_(); m.norm_fn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(3, C0, N)
    ref = m(x)
    torch.testing.assert_close(m.to(DEV)(x.to(DEV)).cpu(), ref, rtol=1e-4, atol=1e-4)
    wide = U.Conv1dNormRelu(300, 280).eval()  # beyond 12 output tiles: fused_mlp1d returns None, library path
    xw = torch.randn(1, 300, 40)
    refw = wide(xw)
    assert U._mlp_pack([wide.to(DEV)]) is None
    torch.testing.assert_close(wide(xw.to(DEV)).cpu(), refw, rtol=1e-4, atol=1e-4)


@torch.no_grad()
def test_pack_rows_concatenates_and_pads():
    """rpe_pointconv_pack_rows: cat([xyz, a, b, c]) channel-last, zero-padded to a multiple of 16, strided sources."""
    torch.manual_seed(11)
    xyz = torch.randn(2, 3, 333, device=DEV)
    a, b, c = torch.randn(2, 64, 333, device=DEV), torch.randn(2, 9, 333, device=DEV)[:, 2:5], torch.randn(2, 1, 333, device=DEV)
    packed = PC.pack_rows(xyz, [a, b, c])
    assert packed.channels == 68 and packed.rows.shape == (2, 333, 80)
    want = torch.cat([xyz, a, b, c], dim=1).transpose(1, 2)
    assert torch.equal(packed.rows[:, :, :71], want) and (packed.rows[:, :, 71:] == 0).all()


@torch.no_grad()
@pytest.mark.parametrize("B,N,C,Cmid,Cout", [(2, 200, 20, 24, 24), (4, 1024, 195, 128, 128), (1, 50, 7, 130, 20), (3, 4096, 35, 32, 32)])
def test_pointconv_rows_output_feeds_the_next_layer(B, N, C, Cmid, Cout):
    """out_rows=True (the fused kernel writes the next layer's gather rows) equals channel-first output + packing pass,
    bit for bit, and the chained second layer equals the two-step form -- every tile configuration the launcher picks."""
    r = I.rng(7000 + N)
    xyz = dev(np.ascontiguousarray(I.ids_cloud(r, B, N).transpose(0, 2, 1)))
    feat = dev(r.standard_normal((B, C, N), dtype=np.float32))
    from rpeflow_amd.csrc import k_nearest_neighbor
    knn = k_nearest_neighbor(xyz, xyz, 16)
    torch.manual_seed(N)
    c1, c2 = PC.PointConvNoSampling(C, Cmid).to(DEV).eval(), PC.PointConvNoSampling(Cmid, Cout).to(DEV).eval()
    y = c1(xyz, feat, knn)
    rows = c1(xyz, feat, knn, out_rows=True)
    again = PC.pack_rows(xyz, y)
    assert torch.equal(rows.rows, again.rows)
    assert torch.equal(c2(xyz, rows, knn), c2(xyz, y, knn))
    ref = O.pointconv({k: v.detach().cpu().numpy() for k, v in c1.state_dict().items()}, xyz.cpu().numpy(), feat.cpu().numpy(),
                      knn_indices=knn.cpu().numpy(), k=16, norm=None)
    close_sum(y, ref, what="first layer vs oracle")


@torch.no_grad()
def test_correlation3d_projection_is_hoistable():
    """Correlation3D.forward(projected=project_stacked(...)) equals the plain call (the model issues the projection early)."""
    c, x = K.BLOCK_CASES["correlation3d"], K.block_inputs("correlation3d")
    m, _ = _load(P3.Correlation3D(c["C"], c["C"], k=c["k"]), _shapes_corr3d(c["C"]), c["seed"] + 1000)
    a = m(dev(x["xyz1"]), dev(x["feat1"]), dev(x["xyz2"]), dev(x["feat2"]))
    proj = m.project_stacked(torch.cat([dev(x["feat1"]), dev(x["feat2"])], 0))
    b = m(dev(x["xyz1"]), dev(x["feat1"]), dev(x["xyz2"]), dev(x["feat2"]), projected=proj)
    assert torch.equal(a, b)


def test_project_feat_takes_the_3d_fusers_samples():
    """project_feat_with_nn_corr(..., sampled_2d=grid_sample_wrapper(feat_2d, xy)) == project_feat_with_nn_corr(...): the 2-D
    fuser reuses what the 3-D fuser of the same (map, points) pair sampled (RPEFlow_core.py:334-337), bit for bit."""
    import rpeflow_amd.utils as U
    g = torch.Generator(device="cpu").manual_seed(5)
    B, C2, C3, H, W, N = 2, 19, 13, 18, 30, 700
    feat_2d = torch.randn(B, C2 + 2, H, W, generator=g).to(DEV)
    feat_3d = torch.randn(B, C3, N, generator=g).to(DEV)
    xy = (torch.rand(B, 2, N, generator=g) * torch.tensor([W + 3.0, H + 3.0])[None, :, None] - 1.5).to(DEV)  # some points outside the map
    nn = torch.randint(0, N, (B, H * W), generator=g).to(DEV)
    plain = U.project_feat_with_nn_corr(xy, feat_2d[:, :C2].contiguous(), feat_3d, nn)
    sampled = U.grid_sample_wrapper(feat_2d, xy)  # a wider map, as the correlation fuser has it: the first C2 channels are a strided view
    shared = U.project_feat_with_nn_corr(xy, feat_2d[:, :C2].contiguous(), feat_3d, nn, sampled_2d=sampled[:, :C2])
    assert torch.equal(plain, shared)


def test_knn_interpolation_with_the_callers_indices():
    g = torch.Generator(device="cpu").manual_seed(6)
    coarse, fine = torch.randn(2, 3, 300, generator=g).to(DEV), torch.randn(2, 3, 700, generator=g).to(DEV)
    feat = torch.randn(2, 11, 300, generator=g).to(DEV)
    plain, idx = U.knn_interpolation(coarse, feat, fine, return_indices=True)
    assert idx.shape == (2, 700, 3) and torch.equal(plain, U.knn_interpolation(coarse, feat, fine))
    assert torch.equal(plain, U.knn_interpolation(coarse, feat, fine, knn_indices=idx))


def test_project_feat_with_the_fusers_two_elementwise_steps_inside():
    """project_feat_with_nn_corr(..., subtract_last=, append=) == cat([project_feat(...) with its last channels minus a map, another
    map]) -- the 2-D correlation fuser's "-= last_flow_2d" and cat with the event features (RPEFlow_core.py:82-83) -- bit for bit
    against the plain call followed by the two PyTorch steps; odd map sizes, with and without caller-provided samples."""
    from rpeflow_amd import utils as U
    torch.manual_seed(11)
    # (the last two: rows too wide for 64 points of them in LDS -> 32 points a workgroup; then too wide for that -> one thread
    # per point and channel slice)
    for B, C2, C3, H, W, N, E in [(2, 81, 34, 9, 15, 256, 32), (1, 32, 5, 7, 5, 40, 3), (3, 16, 66, 18, 30, 512, 64),
                                  (2, 192, 195, 6, 10, 130, 4), (1, 300, 290, 4, 5, 70, 2)]:
        xy = torch.rand(B, 2, N, device="cuda:0") * torch.tensor([W - 1.0, H - 1.0], device="cuda:0").view(1, 2, 1)
        f2, f3 = torch.randn(B, C2, H, W, device="cuda:0"), torch.randn(B, C3, N, device="cuda:0")
        nn = torch.randint(0, N, (B, H * W), device="cuda:0")
        sub, app = torch.randn(B, 2, H, W, device="cuda:0"), torch.randn(B, E, H, W, device="cuda:0")
        plain = U.project_feat_with_nn_corr(xy, f2, f3, nn)
        ref = plain.clone()
        ref[:, -2:] -= sub
        ref = torch.cat([ref, app], dim=1)
        got = U.project_feat_with_nn_corr(xy, f2, f3, nn, subtract_last=sub, append=app)
        assert got.shape == ref.shape and torch.equal(got, ref)
        only_sub = U.project_feat_with_nn_corr(xy, f2, f3, nn, subtract_last=sub)
        assert torch.equal(only_sub, ref[:, :C3 + 3])
        sampled = U.grid_sample_wrapper(f2, xy)
        assert torch.equal(U.project_feat_with_nn_corr(xy, f2, f3, nn, sampled_2d=sampled, subtract_last=sub, append=app), ref)


# ---------------------------------------------------------------- round 5: the glue between the operators inside their launches
@pytest.mark.parametrize("mode", ["parallel", "perspective"])
def test_project_points_equals_the_reference_op_sequence(mode):
    """rpe_project_points == project_pc2image (utils.py:260-285) + the in-place rescale (RPEFlow_core.py:316-324) of both frames,
    computed with the reference's own tensor ops on the CPU: bit for bit (fp32 adds / muls / divs round alike on both sides)."""
    g = torch.Generator().manual_seed(12)
    B, N = 3, 1000
    xyz1 = torch.randn(B, 3, N, generator=g) * 5 + torch.tensor([0.0, 0.0, 30.0])[None, :, None]
    top = torch.randn(2 * B, 3, 2 * N, generator=g) * 5 + torch.tensor([0.0, 0.0, 30.0])[None, :, None]
    xyz2 = top[B:, :, :N]  # a strided prefix view, as the pyramid hands its levels over
    sx, sy = 239 / 29, 143 / 17
    if mode == "parallel":
        cam = {"projection_mode": "parallel", "cx": 14.5, "cy": 8.5}
        ref = [torch.stack([p[:, 0] + cam["cx"], p[:, 1] + cam["cy"]], 1) for p in (xyz1, xyz2)]
    else:
        f, cx, cy = torch.tensor([1050.0, 900.0, 1000.0]), torch.tensor([479.5, 400.0, 500.25]), torch.tensor([269.5, 300.0, 250.75])
        cam = {"projection_mode": "perspective", "f": f, "cx": cx, "cy": cy}
        ref = [torch.stack([cx[:, None] + (f[:, None] / p[:, 2]) * p[:, 0], cy[:, None] + (f[:, None] / p[:, 2]) * p[:, 1]], 1) for p in (xyz1, xyz2)]
    for r in ref:
        r[:, 0] *= sx
        r[:, 1] *= sy
    cam_dev = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in cam.items()}
    got = U.project_points(xyz1.to(DEV), top.to(DEV)[B:, :, :N], cam_dev, sx, sy)
    assert got.shape == (2 * B, 2, N) and torch.equal(got.cpu(), torch.cat(ref, 0))
    assert torch.equal(U.project_points(xyz1.to(DEV), None, cam_dev, sx, sy).cpu(), ref[0])


def test_grid_sample_sources_equals_the_fusers_op_sequence():
    """grid_sample_sources([(corr, -, -), (flow, (sx, sy), flow_3d_xy), (events, -, -)], xy) == what CorrFeatureFuser3D computes
    around its two grid_sample_wrapper calls (RPEFlow_core.py:103-111): flow * scale, cat, sample, -=, sample, cat -- bit for bit
    against those steps through the one-source operator and PyTorch; points outside the map, odd sizes, a channel-sliced source."""
    g = torch.Generator().manual_seed(13)
    for B, C, E, H, W, N in [(2, 81, 32, 18, 30, 512), (1, 5, 3, 7, 9, 100), (3, 81, 128, 9, 15, 256)]:
        corr, flow = torch.randn(B, C, H, W, generator=g).to(DEV), torch.randn(B, 2, H, W, generator=g).to(DEV)
        wide = torch.randn(B, E + 3, H, W, generator=g).to(DEV)
        events = wide[:, 1:E + 1]  # not contiguous: batch stride (E + 3) H W
        xy = (torch.rand(B, 2, N, generator=g) * torch.tensor([W + 3.0, H + 3.0])[None, :, None] - 1.5).to(DEV)
        flow_3d = torch.randn(B, 3, N, generator=g).to(DEV)
        sx, sy = (30 - 1) / (W - 1), (18 - 1) / (H - 1)
        scaled = flow * torch.tensor([sx, sy], device=DEV).view(1, 2, 1, 1)
        a = U.grid_sample_wrapper(torch.cat([corr, scaled], 1), xy)
        a[:, -2:] -= flow_3d[:, :2]
        ref = torch.cat([a, U.grid_sample_wrapper(events.contiguous(), xy)], 1)
        got = U.grid_sample_sources([(corr, None, None), (flow, (sx, sy), flow_3d[:, :2]), (events, None, None)], xy)
        assert got.shape == ref.shape and torch.equal(got, ref)
    # and the one-source form against the oracle (the operator's own parity lives in test_glue_ops_against_*)
    f = torch.randn(2, 7, 6, 11, generator=g)
    xy = torch.rand(2, 2, 50, generator=g) * torch.tensor([10.0, 5.0])[None, :, None]
    close(U.grid_sample_sources([(f.to(DEV), None, None)], xy.to(DEV)), O.grid_sample_wrapper(f.numpy(), xy.numpy()), 3e-6)


def test_knn_interpolation_of_a_pair_and_backwarp_residual():
    """knn_interpolation(xyz, (a, b), q) == knn_interpolation(xyz, cat([a, b]), q), and backwarp_3d's "xyz2 + flow21" inside the
    launch == the separate add: bit for bit; strided sources (channel slices, prefix views)."""
    g = torch.Generator().manual_seed(14)
    coarse, fine = torch.randn(2, 3, 300, generator=g).to(DEV), torch.randn(2, 3, 700, generator=g).to(DEV)
    wide = torch.randn(2, 80, 400, generator=g).to(DEV)
    a, b = wide[:, 2:5, :300], wide[:, 10:74, :300]
    ref = U.knn_interpolation(coarse, torch.cat([a, b], 1), fine)
    got = U.knn_interpolation(coarse, (a, b), fine)
    assert got.shape == (2, 67, 700) and torch.equal(got, ref)
    assert torch.equal(U.knn_interpolation(coarse, (a, wide[:, :0, :300]), fine), ref[:, :3])  # an empty second tensor
    xyz1, xyz2, flow = coarse, torch.randn(2, 3, 300, generator=g).to(DEV), 0.1 * torch.randn(2, 3, 300, generator=g).to(DEV)
    warped = U.backwarp_3d(xyz1, xyz2, flow)
    w = xyz1 + flow
    idx = U.k_nearest_neighbor(w, xyz2, 3)
    assert torch.equal(warped, xyz2 + U._interpolate(w, flow, xyz2, idx, 3, scale=-1.0))
    close(warped, O.backwarp_3d(xyz1.cpu().numpy(), xyz2.cpu().numpy(), flow.cpu().numpy()), 1e-5)


def test_project_feat_with_a_scaled_tail():
    """project_feat_with_nn_corr(..., feat_3d, feat_3d_tail=t, tail_scale=(sx, sy)) == the call on cat([feat_3d, t * (sx, sy)])
    (RPEFlow_core.py:371-373), with and without the caller's samples, on the coarse-map kernel and the per-pixel one."""
    g = torch.Generator().manual_seed(15)
    for B, C2, C3, H, W, N in [(2, 81, 32, 9, 15, 256), (1, 20, 7, 36, 60, 900), (4, 81, 192, 72, 120, 2048)]:
        xy = (torch.rand(B, 2, N, generator=g) * torch.tensor([W - 1.0, H - 1.0])[None, :, None]).to(DEV)
        f2, f3 = torch.randn(B, C2, H, W, generator=g).to(DEV), torch.randn(B, C3, N, generator=g).to(DEV)
        flow = torch.randn(B, 3, N, generator=g).to(DEV)
        nn = torch.randint(0, N, (B, H * W), generator=g).to(DEV)
        sub, app = torch.randn(B, 2, H, W, generator=g).to(DEV), torch.randn(B, 5, H, W, generator=g).to(DEV)
        sx, sy = (W - 1) / 29, (H - 1) / 17
        tail = flow[:, :2] * torch.tensor([sx, sy], device=DEV).view(1, 2, 1)
        ref = U.project_feat_with_nn_corr(xy, f2, torch.cat([f3, tail], 1), nn, subtract_last=sub, append=app)
        got = U.project_feat_with_nn_corr(xy, f2, f3, nn, subtract_last=sub, append=app, feat_3d_tail=flow[:, :2], tail_scale=(sx, sy))
        assert torch.equal(got, ref)
        sampled = U.grid_sample_wrapper(f2, xy)
        assert torch.equal(U.project_feat_with_nn_corr(xy, f2, f3, nn, sampled_2d=sampled, subtract_last=sub, append=app,
                                                       feat_3d_tail=flow[:, :2], tail_scale=(sx, sy)), ref)


def test_flow_unit_conversions_round_as_the_reference_expression():
    """RPEFlow_core.py:363-370 converts flows between feature-map and sensor units with "flow * (image_w - 1) / (sensor_w - 1)":
    a multiply THEN a divide, two fp32 roundings.  The native operators take the conversion as (numerator, denominator) and apply
    fl(fl(x * num) / den) as they read the flow (rpe_scaled): bit for bit the operator on the tensor PyTorch builds from the
    reference's literal expression -- which one multiply by the precomputed ratio is not (counted below, so the test knows it
    distinguishes the two)."""
    g = torch.Generator().manual_seed(17)
    differs = 0
    for B, C, H, W, N, sh, sw in [(2, 81, 72, 120, 2048, 18, 30), (1, 7, 9, 15, 256, 18, 30), (3, 16, 64, 80, 1024, 16, 20)]:
        corr, flow = torch.randn(B, C, H, W, generator=g).to(DEV), (3.0 * torch.randn(B, 2, H, W, generator=g)).to(DEV)
        xy = (torch.rand(B, 2, N, generator=g) * torch.tensor([W - 1.0, H - 1.0])[None, :, None]).to(DEV)
        flow_3d = torch.randn(B, 3, N, generator=g).to(DEV)
        # the 3-D correlation fuser's front (:367-370, :103-111)
        # (the reference's expression on the CPU, where the parity oracle runs: a true division.  PyTorch's GPU kernel for
        # tensor / python-scalar multiplies by the reciprocal instead -- a third rounding pattern, not the reference's)
        cpu = flow.cpu()
        literal = torch.cat([cpu[:, 0:1] * (sw - 1) / (W - 1), cpu[:, 1:2] * (sh - 1) / (H - 1)], dim=1).to(DEV)
        single = flow * torch.tensor([(sw - 1) / (W - 1), (sh - 1) / (H - 1)], device=DEV).view(1, 2, 1, 1)
        differs += int((literal != single).sum())
        ref = U.grid_sample_wrapper(torch.cat([corr, literal], 1), xy)
        ref[:, -2:] -= flow_3d[:, :2]
        got = U.grid_sample_sources([(corr, None, None), (flow, ((sw - 1, W - 1), (sh - 1, H - 1)), flow_3d[:, :2])], xy)
        assert torch.equal(got, ref)
        # the 2-D correlation fuser's projected tail (:363-366, :371-373)
        f3 = torch.randn(B, 12, N, generator=g).to(DEV)
        nn = torch.randint(0, N, (B, H * W), generator=g).to(DEV)
        cpu = flow_3d.cpu()
        tail = torch.cat([cpu[:, 0:1] * (W - 1) / (sw - 1), cpu[:, 1:2] * (H - 1) / (sh - 1)], dim=1).to(DEV)
        ref = U.project_feat_with_nn_corr(xy, corr, torch.cat([f3, tail], 1), nn)
        got = U.project_feat_with_nn_corr(xy, corr, f3, nn, feat_3d_tail=flow_3d[:, :2], tail_scale=((W - 1, sw - 1), (H - 1, sh - 1)))
        assert torch.equal(got, ref)
    assert differs > 0, "no value distinguishes x * num / den from x * (num / den): the test would prove nothing"


def test_pointwise_conv_reads_channel_slices_where_they_lie():
    """conv_module(1x1, x, residual=r) with x and r channel slices of wider tensors (batch stride larger than C P) == the call on
    contiguous copies: the flow heads add the up-sampled flow, a slice of [flow | features] (RPEFlow_core.py:409-410)."""
    torch.manual_seed(16)
    conv = torch.nn.Conv1d(64, 3, 1).to(DEV)
    wide_x, wide_r = torch.randn(4, 70, 512, device=DEV), torch.randn(4, 67, 512, device=DEV)
    x, r = wide_x[:, 3:67], wide_r[:, :3]
    with torch.no_grad():
        got = U.conv_module(conv, x, residual=r)
        assert torch.equal(got, U.conv_module(conv, x.contiguous(), residual=r.contiguous()))
        close(got, (r + conv(x)).cpu().numpy(), 1e-5)
        odd = U.conv_module(conv, wide_x[:, 3:67, :511], residual=wide_r[:, :3, :511])  # not dense per sample: copied, same values
        close(odd, (wide_r[:, :3, :511] + conv(wide_x[:, 3:67, :511])).cpu().numpy(), 1e-5)
        # one sample: PyTorch's batch stride of a size-1 dimension is arbitrary (here (70 * 512, ...) sliced, then a view whose
        # batch stride is 1) -- the kernel must get the dense stride, and a channel slice is written in place where it lies
        one_x, one_r = wide_x[:1, 3:67], wide_r[:1, :3].clone()
        weird = one_x.as_strided(one_x.shape, (1, 512, 1), one_x.storage_offset())
        assert torch.equal(weird, one_x) and weird.stride(0) == 1
        want = U.conv_module(conv, one_x.contiguous(), residual=one_r.contiguous())
        assert torch.equal(U.conv_module(conv, weird, residual=one_r), want)
        wide_one = torch.randn(1, 67, 512, device=DEV)
        slice_r = wide_one[:, 5:8]                      # B == 1: is_contiguous() although it is a slice of a wider tensor
        want = U.conv_module(conv, one_x, residual=slice_r.clone())
        got = U.conv_module(conv, one_x, residual=slice_r, inplace=True)
        assert torch.equal(got, want) and got.data_ptr() == slice_r.data_ptr() and torch.equal(wide_one[:, 5:8], want)


@pytest.mark.parametrize("cin", [1, 64])  # one channel group: the plain kernel; 16 groups on a small map: the K-split form
@pytest.mark.parametrize("act", [None, "relu", "leaky_relu"])
def test_activations_keep_nan_and_signed_zero_like_the_reference(cin, act):
    """act(scale * (w . x) + shift) on NaN, infinities, signed zeros and denormals == the same expression in PyTorch on the CPU,
    bit for bit: nn.ReLU / nn.LeakyReLU(0.1) keep a NaN a NaN and -0 a -0 there (the reference's evaluation masks NaN predictions,
    eval_withocc.py:86-87); fmaxf(u, 0) would return 0 for both."""
    special = np.array([np.nan, -0.0, 0.0, -1.5, 2.5, -np.inf, np.inf, -1e-45, 1e-45, -3e38, 3e38, 1.0], dtype=np.float32)
    x = np.zeros((1, cin, 128), dtype=np.float32)
    x[0, 0] = np.tile(special, 11)[:128]
    w = np.zeros((1, cin, 1), dtype=np.float32)
    w[0, 0, 0] = 1.0
    fn = {None: lambda t: t, "relu": torch.relu, "leaky_relu": lambda t: torch.nn.functional.leaky_relu(t, 0.1)}[act]
    for scale, shift in ((1.0, 0.0), (-1.0, -0.0), (0.5, 1.0)):  # (-1, -0): zeros come out as -0 in front of the activation
        sc, sh = torch.tensor([scale]), torch.tensor([shift])
        want = fn(torch.nn.functional.conv1d(torch.from_numpy(x), torch.from_numpy(w)) * sc.view(1, -1, 1) + sh.view(1, -1, 1))
        got = U.pointwise_conv(dev(x), dev(w), None, 1, epilogue=(sc.to(DEV), sh.to(DEV), act))
        bits = lambda t: t.detach().cpu().contiguous().numpy().view(np.uint32)
        nan = torch.isnan(want).numpy()
        assert np.array_equal(np.isnan(got.cpu().numpy()), nan)
        assert np.array_equal(bits(got)[~nan], bits(want)[~nan]), (act, scale, shift)


def test_pyramid_of_stacked_clouds_and_constant_level0_feature():
    """build_pc_pyramid on the two halves of one [2B,3,N] tensor (no cat, one gather) == on separate tensors; FeaturePyramid3D's
    level-0 feature computed on one point and broadcast == computed on every point (pwc3d_core.py:51-52), bit for bit."""
    g = torch.Generator().manual_seed(17)
    both = (torch.randn(4, 3, 2000, generator=g) * 5).to(DEV)
    a, b = both[:2].clone(), both[2:].clone()
    x1, x2, i1, i2, stacked = P3.build_pc_pyramid(both[:2], both[2:], [512, 128], return_both=True)
    y1, y2, j1, j2 = P3.build_pc_pyramid(a, b, [512, 128])
    for u, v in zip(x1 + x2 + i1 + i2, y1 + y2 + j1 + j2):
        assert torch.equal(u, v)
    for lvl in range(3):
        assert torch.equal(stacked[lvl], torch.cat([x1[lvl], x2[lvl]], 0))
    torch.manual_seed(18)
    pyramid = P3.FeaturePyramid3D([16, 32, 64], norm="batch_norm", k=16).to(DEV).eval()
    for m in pyramid.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_()
            m.running_var.uniform_(0.5, 2.0)
    with torch.no_grad():
        feats = pyramid(stacked)
        full = pyramid.level0_mlp(torch.zeros_like(stacked[0]))
        assert feats[0].shape == full.shape and torch.equal(feats[0], full)
        assert feats[0].stride(2) == 0  # ONE vector, broadcast
    with torch.enable_grad():
        assert pyramid(stacked)[0].stride(2) != 0  # (with autograd on: the reference's op sequence)
