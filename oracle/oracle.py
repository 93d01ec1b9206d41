"""CPU oracle for the RPEFlow hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``rpeflow_amd/`` does, and the
product path raises when its HIP library is missing instead of falling back
to anything here.

Each function restates, on numpy arrays, one function of the reference's
CPU/PyTorch path (file:line relative to /root/reference).  The bit-exact ops
(squared_distance, k_nearest_neighbor, furthest_point_sampling) and the
samplers run in ``rpe_oracle.c`` with every rounding written out; the
compositions around them are numpy float32.

Parity pin: ``tests/golden/*.npz`` hold outputs of the imported reference,
generated in the build container by ``tests/golden/make_golden.py``;
``tests/test_oracle_golden.py`` checks this module against them.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librpe_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile rpe_oracle.c with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "rpe_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_selfcheck.restype = ctypes.c_int
        _lib.orc_knn.restype = ctypes.c_int
        if _lib.orc_selfcheck() != 0:
            raise RuntimeError("oracle: fmaf() is not a single-rounding fma on this host")
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_f32p)


# --------------------------------------------------------------------------
# the four operators of models/csrc/__init__.py:1
# --------------------------------------------------------------------------
def squared_distance(xyz1, xyz2):
    """models/csrc/wrapper.py:40-52.  xyz1 [B,N1,D], xyz2 [B,N2,D], D<=3."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    assert xyz1.shape[-1] == xyz2.shape[-1] and xyz1.shape[-1] <= 3
    B, N1, D = xyz1.shape
    N2 = xyz2.shape[1]
    out = np.empty((B, N1, N2), np.float32)
    lib().orc_squared_distance(_p(xyz1), _p(xyz2), B, N1, N2, D, _p(out))
    return out


def k_nearest_neighbor(input_xyz, query_xyz, k, return_dists=False, ties="torch"):
    """models/csrc/wrapper.py:106-127 (CPU branch :115-117), including the
    ``shape[1] <= 3`` channel-first sniff (:119-122).  ties="torch": equal distances selected and ordered as
    torch.topk does on the CPU (libstdc++ partial_sort / nth_element / sort restated in rpe_oracle.c);
    ties="index": lower index first."""
    input_xyz, query_xyz = np.asarray(input_xyz), np.asarray(query_xyz)
    if input_xyz.shape[1] <= 3:
        assert query_xyz.shape[1] == input_xyz.shape[1]
        input_xyz = input_xyz.transpose(0, 2, 1)
        query_xyz = query_xyz.transpose(0, 2, 1)
    input_xyz, query_xyz = _f32(input_xyz), _f32(query_xyz)
    B, M, D = input_xyz.shape
    Q = query_xyz.shape[1]
    if k > M:
        raise RuntimeError("selected index k out of range")  # what torch.topk raises
    idx = np.empty((B, Q, k), np.int64)
    dist = np.empty((B, Q, k), np.float32)
    fn = lib().orc_knn_torch_ties if ties == "torch" else lib().orc_knn
    rc = fn(_p(input_xyz), _p(query_xyz), B, M, Q, D, k, idx.ctypes.data_as(_i64p), _p(dist))
    assert rc == 0
    return (idx, dist) if return_dists else idx


def furthest_point_sampling(xyz, n_samples):
    """models/csrc/wrapper.py:75-103 (CPU branch :83-96).  xyz [B,N,3]."""
    xyz = _f32(xyz)
    assert xyz.shape[2] == 3 and xyz.shape[1] > n_samples
    B, N, _ = xyz.shape
    idx = np.empty((B, n_samples), np.int64)
    lib().orc_fps(_p(xyz), B, N, n_samples, idx.ctypes.data_as(_i64p))
    return idx


def correlation2d(input1, input2, max_displacement):
    """models/csrc/wrapper.py:55-72 (_correlation_py :56-65).  NCHW in/out."""
    input1, input2 = _f32(input1), _f32(input2)
    B, C, H, W = input1.shape
    n = 2 * max_displacement + 1
    out = np.empty((B, n * n, H, W), np.float32)
    lib().orc_correlation2d(_p(input1), _p(input2), B, C, H, W, max_displacement, _p(out))
    return out


def correlation2d_backward(grad_output, input1, input2, max_displacement):
    """Gradients of _correlation_py (wrapper.py:56-65) w.r.t. both inputs, as autograd derives them: every cost plane
    k = mean_c(in1 * shift_k(in2)) sends grad/C * shift_k(in2) to in1 and the un-shifted grad/C * in1 to in2.
    float64 accumulation, rounded once.  NCHW."""
    go, a, b = np.asarray(grad_output, np.float64), np.asarray(input1, np.float64), np.asarray(input2, np.float64)
    B, C, H, W = a.shape
    md, n = max_displacement, 2 * max_displacement + 1
    bp = np.pad(b, ((0, 0), (0, 0), (md, md), (md, md)))
    g1, g2p = np.zeros_like(a), np.zeros_like(bp)
    for i in range(n):
        for j in range(n):
            g = go[:, i * n + j][:, None] / C
            g1 += g * bp[:, :, i:i + H, j:j + W]
            g2p[:, :, i:i + H, j:j + W] += g * a
    return g1.astype(np.float32), g2p[:, :, md:md + H, md:md + W].astype(np.float32)


# --------------------------------------------------------------------------
# glue ops of models/utils.py
# --------------------------------------------------------------------------
def batch_indexing_channel_first(data, indices):
    """models/utils.py:119-137.  data [B,C,N], indices [B,...] -> [B,C,...]."""
    data, indices = np.asarray(data), np.asarray(indices)
    B, C = data.shape[:2]
    flat = indices.reshape(B, 1, -1).astype(np.int64)
    out = np.take_along_axis(data, np.broadcast_to(flat, (B, C, flat.shape[2])), axis=2)
    return out.reshape((B, C) + indices.shape[1:])


def batch_indexing_channel_last(data, indices):
    """models/utils.py:101-116.  data [B,N,C] (or [B,N]), indices [B,...]."""
    data, indices = np.asarray(data), np.asarray(indices).astype(np.int64)
    B = data.shape[0]
    bidx = np.arange(B).reshape((B,) + (1,) * (indices.ndim - 1))
    return data[bidx, indices]


def _bilinear(feat, px, py, border):
    feat, px, py = _f32(feat), _f32(px), _f32(py)
    B, C, H, W = feat.shape
    P = px.shape[1]
    out = np.empty((B, C, P), np.float32)
    lib().orc_bilinear_sample(_p(feat), B, C, H, W, _p(px), _p(py), P, int(border), _p(out))
    return out


def backwarp_2d(x, flow12, padding_mode="border"):
    """models/utils.py:186-198 (+ mesh_grid :172-183).  x [B,C,H,W], flow [B,2,H,W]."""
    assert padding_mode in ("border", "zeros")
    x, flow12 = _f32(x), _f32(flow12)
    B, C, H, W = x.shape
    gx = np.arange(W, dtype=np.float32)[None, None, :] + flow12[:, 0]
    gy = np.arange(H, dtype=np.float32)[None, :, None] + flow12[:, 1]
    out = _bilinear(x, gx.reshape(B, -1), gy.reshape(B, -1), padding_mode == "border")
    return out.reshape(B, C, H, W)


def grid_sample_wrapper(feat_2d, xy):
    """models/utils.py:288-294.  feat_2d [B,C,H,W], xy [B,2,N] -> [B,C,N]; padding 'zeros'."""
    xy = _f32(xy)
    return _bilinear(feat_2d, xy[:, 0], xy[:, 1], False)


def knn_interpolation(input_xyz, input_features, query_xyz, k=3):
    """models/utils.py:140-156.  channel-first [B,3,M], [B,C,M], [B,3,Q] -> [B,C,Q]."""
    input_xyz, input_features, query_xyz = _f32(input_xyz), _f32(input_features), _f32(query_xyz)
    knn = k_nearest_neighbor(input_xyz, query_xyz, k)
    knn_xyz = batch_indexing_channel_first(input_xyz, knn)
    diff = knn_xyz - query_xyz[..., None]
    dists = np.sqrt(np.sum(diff * diff, axis=1, dtype=np.float32)).astype(np.float32)
    dists = np.maximum(dists, np.float32(1e-8))
    w = (np.float32(1.0) / dists).astype(np.float32)
    w = w / np.sum(w, axis=-1, keepdims=True, dtype=np.float32)
    feats = batch_indexing_channel_first(input_features, knn)
    return np.sum(feats * w[:, None], axis=-1, dtype=np.float32)


def backwarp_3d(xyz1, xyz2, flow12, k=3):
    """models/utils.py:159-169."""
    xyz1, xyz2, flow12 = _f32(xyz1), _f32(xyz2), _f32(flow12)
    flow21 = knn_interpolation(xyz1 + flow12, -flow12, xyz2, k)
    return xyz2 + flow21


def project_feat_with_nn_corr(xy, feat_2d, feat_3d, nn_indices=None):
    """models/utils.py:297-317.  xy [B,2,N], feat_2d [B,C2,H,W], feat_3d [B,C3,N],
    nn_indices [B,H*W] -> [B,C3+3,H,W]."""
    xy, feat_2d, feat_3d = _f32(xy), _f32(feat_2d), _f32(feat_3d)
    B, C2, H, W = feat_2d.shape
    gx = np.broadcast_to(np.arange(W, dtype=np.float32)[None, :], (H, W)).reshape(-1)
    gy = np.broadcast_to(np.arange(H, dtype=np.float32)[:, None], (H, W)).reshape(-1)
    grid = np.broadcast_to(np.stack([gx, gy])[None], (B, 2, H * W))
    if nn_indices is None:
        nn_indices = k_nearest_neighbor(xy, grid, 1)[..., 0]
    nn_feat2d = batch_indexing_channel_first(grid_sample_wrapper(feat_2d, xy), nn_indices)
    nn_feat3d = batch_indexing_channel_first(feat_3d, nn_indices)
    nn_offset = batch_indexing_channel_first(xy, nn_indices) - grid
    nn_corr = np.mean(nn_feat2d * feat_2d.reshape(B, C2, H * W), axis=1, keepdims=True, dtype=np.float32)
    out = np.concatenate([nn_offset, nn_corr, nn_feat3d], axis=1)
    return out.reshape(B, -1, H, W).astype(np.float32)


# --------------------------------------------------------------------------
# 3D blocks: models/pointconv.py, models/pwc3d_core.py.  Parameters arrive as
# a dict of numpy arrays keyed like the module's state_dict.
# --------------------------------------------------------------------------
def _leaky(x, slope):
    return np.where(x >= 0, x, x * np.float32(slope)).astype(np.float32)


def _mlp_1x1(x, params, prefix, n_layers, act):
    """MLP1d/MLP2d of 1x1 convs, norm=None: models/utils.py:65-98.  x [B,C,...]."""
    for i in range(n_layers):
        w = params[f"{prefix}.convs.{i}.conv_fn.weight"].reshape(-1, x.shape[1]).astype(np.float32)
        b = params[f"{prefix}.convs.{i}.conv_fn.bias"].astype(np.float32)
        x = np.einsum("oc,bc...->bo...", w, x, dtype=np.float32) + b.reshape((1, -1) + (1,) * (x.ndim - 2))
        x = _leaky(x, 0.1) if act == "leaky_relu" else np.maximum(x, 0).astype(np.float32)
    return x


def pointconv(params, xyz, features, sampled_xyz=None, knn_indices=None, k=16, norm=None, eps=1e-5):
    """PointConvDownSampling.forward (models/pointconv.py:33-61) when ``sampled_xyz``
    is given, PointConvNoSampling.forward (:90-122) otherwise.  activation leaky(0.1).
    norm: None or 'batch_norm' (eval mode, running stats from params)."""
    xyz, features = _f32(xyz), _f32(features)
    q_xyz = xyz if sampled_xyz is None else _f32(sampled_xyz)
    B = xyz.shape[0]
    feats = np.concatenate([xyz, features], axis=1)  # [B, C+3, N]
    if knn_indices is None:
        knn_indices = k_nearest_neighbor(xyz, q_xyz, k)
    else:
        knn_indices = np.asarray(knn_indices)[:, :, :k]
    knn_xyz = batch_indexing_channel_first(xyz, knn_indices)  # [B,3,Q,k]
    rel = knn_xyz - q_xyz[:, :, :, None]
    w = _mlp_1x1(rel, params, "weight_net", 2, "leaky_relu")  # [B,16,Q,k]
    w = w.transpose(0, 2, 1, 3)  # [B,Q,16,k]
    knn_feat = batch_indexing_channel_last(feats.transpose(0, 2, 1), knn_indices)  # [B,Q,k,C+3]
    wf = np.matmul(w, knn_feat).reshape(B, q_xyz.shape[2], -1)  # [B,Q,16*(C+3)]
    out = wf @ params["linear.weight"].T.astype(np.float32) + params["linear.bias"].astype(np.float32)
    out = out.transpose(0, 2, 1)  # [B,Cout,Q]
    if norm == "batch_norm":
        g, b_ = params["norm_fn.weight"], params["norm_fn.bias"]
        m, v = params["norm_fn.running_mean"], params["norm_fn.running_var"]
        out = (out - m[None, :, None]) / np.sqrt(v[None, :, None] + np.float32(eps)) * g[None, :, None] + b_[None, :, None]
    return _leaky(out.astype(np.float32), 0.1)


def correlation3d(params, xyz1, feat1, xyz2, feat2, knn_indices_1in1=None, k=16):
    """Correlation3D.forward: models/pwc3d_core.py:69-117."""
    xyz1, feat1, xyz2, feat2 = _f32(xyz1), _f32(feat1), _f32(xyz2), _f32(feat2)
    B, C, N = feat1.shape
    knn12 = k_nearest_neighbor(xyz2, xyz1, k)
    rel2 = batch_indexing_channel_first(xyz2, knn12) - xyz1[:, :, :, None]
    f2 = batch_indexing_channel_first(feat2, knn12)
    f1 = np.broadcast_to(feat1[:, :, :, None], (B, C, N, k))
    cat = np.concatenate([f1, f2, rel2], axis=1)
    p2p = _mlp_1x1(cat, params, "cost_mlp", 2, "leaky_relu")
    w2 = _mlp_1x1(rel2, params, "weight_net2", 3, "relu")
    p2n = np.sum(w2 * p2p, axis=3, dtype=np.float32)
    if knn_indices_1in1 is None:
        knn_indices_1in1 = k_nearest_neighbor(xyz1, xyz1, k)
    rel1 = batch_indexing_channel_first(xyz1, knn_indices_1in1) - xyz1[:, :, :, None]
    w1 = _mlp_1x1(rel1, params, "weight_net1", 3, "relu")
    n2n = batch_indexing_channel_first(p2n, knn_indices_1in1)
    return np.sum(w1 * n2n, axis=3, dtype=np.float32)


# --------------------------------------------------------------------------
# event voxelisation (event_utils.py:109-128, 211-303; temporal_bilinear=True)
# --------------------------------------------------------------------------
def events_to_voxel(events, num_bins, height, width, event_polarity):
    """events [N,4] (x, y, t, polarity), float64 or float32.  Restates eventsToXYTP(post_process=True) -> events_to_voxel_torch /
    events_to_neg_pos_voxel_torch in the arithmetic numpy and torch use for the array's dtype: float64 arrays in float64; float32
    arrays (what load_events_h5 returns, event_utils.py:11-20) in float32 throughout -- deltaT + 1e-6 stays a float32 under
    numpy 2's scalar promotion, which is what the goldens were generated with.  Weights rounded to float32, accumulated per
    bin in event order (np.add.at is sequential, like index_put_(accumulate=True) on the CPU)."""
    ev = np.asarray(events)
    T = np.float32 if ev.dtype == np.float32 else np.float64
    ev = ev.astype(T, copy=False)
    xs, ys, ps = ev[:, 0].astype(np.int32), ev[:, 1].astype(np.int32), ev[:, 3].astype(np.int32)
    ts = ((ev[:, 2] - ev[0, 2]) / T((ev[-1, 2] - ev[0, 2]) + T(1e-6))).astype(T)
    with np.errstate(invalid="ignore", divide="ignore"):  # one timestamp for every event: 0 / 0, NaN weights, as in the reference
        t_norm = ((ts - ts[0]) / T(ts[-1] - ts[0]) * T(num_bins - 1)).astype(T)
    grids = [np.where(ps > 0, T(1), T(0)), np.where(ps <= 0, T(1), T(0))] if event_polarity else [ps.astype(T)]
    out = []
    for wgt in grids:
        for b in range(num_bins):
            img = np.zeros((height, width), np.float32)
            np.add.at(img, (ys, xs), (wgt * np.maximum(T(0), T(1) - np.abs(t_norm - T(b)))).astype(np.float32))
            out.append(img)
    return np.stack(out)


# --------------------------------------------------------------------------
# IDS transforms (models/utils.py:320-377).  fp32 numpy arithmetic rounds every operation like the reference's
# CPU tensors do; the logarithm / exponential are the correctly rounded fp32 ones (float64 function, rounded once).
# The reference's own CPU log (MKL high-accuracy vsLn behind torch.log) is correctly rounded on all but ~2e-4 of
# its inputs and differs between CPU models; tests/test_oracle_golden.py bounds the difference against the clouds
# the reference produced in the build container.
def _ids_scales(H, W, Hp, Wp):
    sw, sh = (Wp - 1) / (W - 1), (Hp - 1) / (H - 1)
    return [np.float32(v) for v in (sw, sh, (Wp - 1) / 2, (Hp - 1) / 2, min(sw, sh))]


def perspect2parallel(xyz, intrinsics, H, W, Hp, Wp):
    """utils.py:320-346.  xyz [B,3,N], intrinsics [B,3] = (f, cx, cy); sensor H x W -> parallel sensor Hp x Wp."""
    xyz, intr = _f32(xyz), _f32(intrinsics)
    sw, sh, hw, hh, sz = _ids_scales(H, W, Hp, Wp)
    f, cx, cy = intr[:, 0:1], intr[:, 1:2], intr[:, 2:3]
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    fz = f / z
    log_z = np.log(z.astype(np.float64)).astype(np.float32)
    return np.stack([(cx + fz * x) * sw - hw, (cy + fz * y) * sh - hh, (f * log_z + np.float32(1)) * sz], axis=1).astype(np.float32)


def parallel2perspect(xyz, intrinsics, H, W, Hp, Wp):
    """utils.py:349-377."""
    xyz, intr = _f32(xyz), _f32(intrinsics)
    sw, sh, hw, hh, sz = _ids_scales(H, W, Hp, Wp)
    f, cx, cy = intr[:, 0:1], intr[:, 1:2], intr[:, 2:3]
    x, y, z = (xyz[:, 0] + hw) / sw, (xyz[:, 1] + hh) / sh, xyz[:, 2] / sz
    oz = np.exp(((z - np.float32(1)) / f).astype(np.float64)).astype(np.float32)
    return np.stack([(x - cx) * oz / f, (y - cy) * oz / f, oz], axis=1).astype(np.float32)
