"""PointConv layers -- mirror of the reference's models/pointconv.py.

Same constructor arguments, forward signatures and state-dict keys
(``weight_net.convs.{0,1}.conv_fn.*``, ``linear.*``, ``norm_fn.*``), so the
reference's checkpoints load.  forward() runs the KNN kernel, ONE fused
grouping kernel (gather + weight net + k-reduction, csrc/pointconv.hip) and
the nn.Linear on hipBLASLt; the reference's six intermediate tensors
(pointconv.py:49-57) are never materialised.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .csrc import k_nearest_neighbor
from .utils import MLP2d, _act, _norm, affine_epilogue


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def pointconv_group(xyz, features, query_xyz, knn_indices, weight_net, slope=0.1):
    """[B,3,M], [B,C,M], [B,3,Q], [B,Q,>=16] -> [B,Q,16*(C+3)] (pointconv.py:43-57)."""
    _lib.require_gpu(xyz, features, query_xyz, knn_indices, op="pointconv_group")
    B, _, M = xyz.shape
    Q = query_xyz.shape[2]
    feats_cl = torch.cat([xyz, features], dim=1).transpose(1, 2).contiguous().float()  # [B,M,C+3]
    CF = feats_cl.shape[2]
    c0, c1 = weight_net.convs[0].conv_fn, weight_net.convs[1].conv_fn
    w1, b1 = c0.weight.detach().reshape(8, 3).contiguous().float(), c0.bias.detach().contiguous().float()
    w2, b2 = c1.weight.detach().reshape(16, 8).contiguous().float(), c1.bias.detach().contiguous().float()
    knn_indices = knn_indices.to(torch.int64)
    if knn_indices.stride(2) != 1 or knn_indices.stride(0) != Q * knn_indices.stride(1):
        knn_indices = knn_indices.contiguous()
    xyz, query_xyz = xyz.float(), query_xyz.float()
    out = torch.empty((B, Q, 16 * CF), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        rc = _lib.lib().rpe_pointconv_group(
            _ptr(xyz), *xyz.stride(), _ptr(query_xyz), *query_xyz.stride(), _ptr(feats_cl),
            _ptr(knn_indices), knn_indices.stride(1), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), float(slope),
            B, M, Q, CF, _ptr(out), _lib.stream_of(xyz))
    _lib.check(rc, "pointconv_group")
    return out


class _PointConv(nn.Module):
    def __init__(self, in_channels, out_channels, norm=None, activation="leaky_relu", k=16):
        super().__init__()
        if k != 16:
            raise NotImplementedError("rpeflow_amd PointConv is built for k=16 (conf/*/*.yaml pwc3d.k)")
        if activation != "leaky_relu":
            raise NotImplementedError("rpeflow_amd PointConv fuses the weight net's leaky_relu(0.1)")
        self.k = k
        self.weight_net = MLP2d(3, [8, 16], activation=activation)  # pointconv.py:12
        self.linear = nn.Linear(16 * (in_channels + 3), out_channels)  # pointconv.py:13
        self.norm_fn = _norm(norm, out_channels, 1)
        self.activation_fn = _act(activation)

    def _finish(self, grouped):
        epi = affine_epilogue(self, self.linear.bias, self.norm_fn, self.activation_fn) if grouped.is_cuda else None
        if epi is None:
            out = self.linear(grouped).float()  # [B,Q,Cout]
            return self.activation_fn(self.norm_fn(out.transpose(1, 2)))
        # W [Cout,K] x grouped^T [B,K,Q] lands channel-first with no transpose pass; bias + eval BatchNorm + activation
        # follow as one in-place kernel
        from .restormer_ops import channel_affine_act_
        scale, shift, kind = epi
        out = torch.matmul(self.linear.weight, grouped.transpose(1, 2))  # [B,Cout,Q]
        return channel_affine_act_(out, scale, shift, kind, 0.1)


class PointConvDownSampling(_PointConv):
    """pointconv.py:7-61."""

    def forward(self, xyz, features, sampled_xyz, knn_indices=None):
        """``knn_indices`` [B,Q,k]: neighbours of sampled_xyz in xyz if the caller already has them (the reference
        always searches here, pointconv.py:46)."""
        if knn_indices is None:
            knn_indices = k_nearest_neighbor(xyz, sampled_xyz, self.k)  # [B,Q,k]
        return self._finish(pointconv_group(xyz, features, sampled_xyz, knn_indices, self.weight_net))


class PointConvNoSampling(_PointConv):
    """pointconv.py:64-122."""

    def forward(self, xyz, features, knn_indices=None):
        batch_size, n_points = xyz.shape[0], xyz.shape[2]
        if knn_indices is not None:
            assert knn_indices.shape[:2] == torch.Size([batch_size, n_points])
            assert knn_indices.shape[2] >= self.k
        else:
            knn_indices = k_nearest_neighbor(xyz, xyz, self.k)
        return self._finish(pointconv_group(xyz, features, xyz, knn_indices, self.weight_net))
