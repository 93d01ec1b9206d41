"""PointConv layers -- mirror of the reference's models/pointconv.py.

Same constructor arguments, forward signatures and state-dict keys
(``weight_net.convs.{0,1}.conv_fn.*``, ``linear.*``, ``norm_fn.*``), so the
reference's checkpoints load.  forward() is the KNN kernel (unless the caller
brings the neighbour table), one pass that lays ``cat([xyz, features])`` out
channel-last (pointconv.py:43-44) and ONE fused kernel for everything else --
gather, weight net, the 16 x (C+3) weighted sums, nn.Linear, bias, eval-mode
BatchNorm and the activation (csrc/pointconv_fused.hip).  None of the
reference's intermediates (pointconv.py:49-57; [B,Q,16,C+3] and [B,Q,16(C+3)]
are 208 MB each at pyramid level 1) is ever written to memory.

The fused kernel is built for what every shipped configuration uses (k = 16,
leaky_relu, no norm or eval-mode BatchNorm, inference).  Everything else the
reference's constructor accepts -- any k, ``instance_norm``, ``relu`` / no
activation, BatchNorm in training mode, inputs that require grad -- takes the
reference's own op sequence (pointconv.py:43-58) on the GPU: the HIP neighbour
search and gathers, library GEMMs, the norm / activation modules themselves.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .csrc import k_nearest_neighbor
from .utils import MLP2d, _act, _inference_only, _norm, affine_epilogue, batch_indexing_channel_first, batch_indexing_channel_last

_ACT = {None: 0, "relu": 1, "leaky_relu": 2}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class PackedRows:
    """``cat([xyz, features])`` channel-last, [B,M,CFp] with CFp = 16 * ceil((C+3)/16), zero beyond column C+3: what the
    fused kernel gathers from.  Produced by pack_rows() or by a PointConv layer asked for ``out_rows=True``."""

    def __init__(self, rows, channels):
        self.rows, self.channels = rows, channels  # channels = C (features only)


def pack_rows(xyz, features):
    """[B,3,M] and [B,C,M] (or a list of up to four such tensors, concatenated along the channels) -> PackedRows."""
    sources = list(features) if isinstance(features, (list, tuple)) else [features]
    _lib.require_gpu(xyz, *sources, op="pointconv pack_rows")
    assert 1 <= len(sources) <= 4
    xyz = xyz if xyz.dtype == torch.float32 else xyz.float()
    sources = [s if s.dtype == torch.float32 else s.float() for s in sources]
    B, _, M = xyz.shape
    C = sum(s.shape[1] for s in sources)
    CFp = (C + 3 + 15) // 16 * 16
    rows = torch.empty((B, M, CFp), dtype=torch.float32, device=xyz.device)
    n = len(sources)
    ptrs = (ctypes.c_void_p * n)(*[s.data_ptr() for s in sources])
    strides = (ctypes.c_int64 * (3 * n))(*[v for s in sources for v in s.stride()])
    chans = (ctypes.c_int * n)(*[s.shape[1] for s in sources])
    with torch.cuda.device(xyz.device):
        rc = _lib.lib().rpe_pointconv_pack_rows(_ptr(xyz), *xyz.stride(), ptrs, strides, chans, n, B, M, CFp, _ptr(rows),
                                                _lib.stream_of(xyz))
    _lib.check(rc, "pointconv pack_rows")
    return PackedRows(rows, C)


def pack_linear(weight, C):
    """nn.Linear(16*(C+3), Cout).weight -> the fused kernel's B-operand fragments (include/rpeflow_hip.h):
    packed[ci][g][t][kk][n][s] = W[16t + n][(4kk + s)*(C+3) + 16ci + g], zero-padded; returns (tensor, n_tiles)."""
    Cout, CF = weight.shape[0], C + 3
    assert weight.shape[1] == 16 * CF
    CFp, n_tiles = (CF + 15) // 16 * 16, (Cout + 127) // 128 * 8
    w = torch.zeros((16 * n_tiles, 16, CFp), dtype=torch.float32, device=weight.device)
    w[:Cout, :, :CF] = weight.detach().float().reshape(Cout, 16, CF)
    #     o = (t, n)      w = (kk, s)      c = (ci, g)
    w = w.reshape(n_tiles, 16, 4, 4, CFp // 16, 16).permute(4, 5, 0, 2, 1, 3).contiguous()
    return w, n_tiles


def gather_channel_first(data, idx):
    """batch_indexing_channel_first (utils.py:119-137): the HIP gather, or torch.gather where a gradient has to flow."""
    if torch.is_grad_enabled() and data.requires_grad:
        B, C = data.shape[:2]
        return torch.gather(data, 2, idx.reshape(B, 1, -1).expand(-1, C, -1)).view([B, C] + list(idx.shape[1:]))
    return batch_indexing_channel_first(data, idx)


def gather_channel_last(data, idx):
    """batch_indexing_channel_last (utils.py:101-116), likewise."""
    if torch.is_grad_enabled() and data.requires_grad:
        B, _, C = data.shape
        return torch.gather(data, 1, idx.reshape(B, -1, 1).expand(-1, -1, C)).view([B] + list(idx.shape[1:]) + [C])
    return batch_indexing_channel_last(data, idx)


class _PointConv(nn.Module):
    def __init__(self, in_channels, out_channels, norm=None, activation="leaky_relu", k=16):
        super().__init__()
        self.k = k
        self.fusable = k == 16 and activation == "leaky_relu" and norm in (None, "batch_norm")  # what pointconv_fused.hip is built for
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight_net = MLP2d(3, [8, 16], activation=activation)  # pointconv.py:12
        self.linear = nn.Linear(16 * (in_channels + 3), out_channels)  # pointconv.py:13
        self.norm_fn = _norm(norm, out_channels, 1)
        self.activation_fn = _act(activation)
        self._packed = None

    def _weights(self):
        """Kernel-layout copies of the parameters; rebuilt when one changes (load_state_dict, .to(), an optimiser step)."""
        params = [self.linear.weight] + [t for conv in self.weight_net.convs for t in (conv.conv_fn.weight, conv.conv_fn.bias)]
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._packed is None or self._packed[0] != key:
            c0, c1 = self.weight_net.convs[0].conv_fn, self.weight_net.convs[1].conv_fn
            lp, n_tiles = pack_linear(self.linear.weight, self.in_channels)
            self._packed = (key, dict(
                w1=c0.weight.detach().reshape(8, 3).contiguous().float(), b1=c0.bias.detach().contiguous().float(),
                w2=c1.weight.detach().reshape(16, 8).contiguous().float(), b2=c1.bias.detach().contiguous().float(),
                lp=lp, n_tiles=n_tiles))
        return self._packed[1]

    def _general(self, xyz, features, query_xyz, knn_indices, out_rows):
        """pointconv.py:43-58 op for op (any k / norm / activation, training-mode norms, autograd)."""
        if isinstance(features, PackedRows):
            features = features.rows[:, :, 3:3 + features.channels].transpose(1, 2)
        elif isinstance(features, (list, tuple)):
            features = torch.cat(list(features), dim=1)
        batch_size, n_samples = xyz.shape[0], query_xyz.shape[2]
        feats_cl = torch.cat([xyz, features], dim=1).transpose(1, 2)            # [B, M, C + 3]
        knn_indices = knn_indices[:, :, :self.k]
        knn_xyz_norm = gather_channel_first(xyz, knn_indices) - query_xyz[:, :, :, None]   # [B, 3, Q, k]
        weights = self.weight_net(knn_xyz_norm).transpose(1, 2)                # [B, Q, 16, k]
        knn_features = gather_channel_last(feats_cl, knn_indices)              # [B, Q, k, C + 3]
        weighted = torch.matmul(weights, knn_features).reshape(batch_size, n_samples, -1)
        out = self.activation_fn(self.norm_fn(self.linear(weighted).float().transpose(1, 2)))
        return pack_rows(query_xyz, out) if out_rows else out

    def _forward(self, xyz, features, query_xyz, knn_indices, out_rows):
        epi = affine_epilogue(self, self.linear.bias, self.norm_fn, self.activation_fn) if self.fusable else None
        raw = [features.rows] if isinstance(features, PackedRows) else list(features) if isinstance(features, (list, tuple)) else [features]
        if epi is None or not _inference_only(xyz, query_xyz, *raw, *self.parameters()):
            return self._general(xyz, features, query_xyz, knn_indices, out_rows)
        packed = features if isinstance(features, PackedRows) else pack_rows(xyz, features)
        return self._run(packed, query_xyz, knn_indices, out_rows, epi)

    def _run(self, packed, query_xyz, knn_indices, out_rows, epi):
        rows = packed.rows
        _lib.require_gpu(rows, query_xyz, knn_indices, op="PointConv")
        if packed.channels != self.in_channels:
            raise RuntimeError("PointConv: %d feature channels given, layer built for %d" % (packed.channels, self.in_channels))
        scale, shift, kind = epi
        w = self._weights()
        B, M, CFp = rows.shape
        Q = query_xyz.shape[2]
        knn_indices = knn_indices.to(torch.int64)
        if knn_indices.stride(2) != 1 or knn_indices.stride(0) != Q * knn_indices.stride(1):
            knn_indices = knn_indices.contiguous()
        query_xyz = query_xyz if query_xyz.dtype == torch.float32 else query_xyz.float()
        Cout = self.out_channels
        if out_rows:
            stride = (Cout + 3 + 15) // 16 * 16
            out = torch.empty((B, Q, stride), dtype=torch.float32, device=rows.device)
        else:
            stride = 0
            out = torch.empty((B, Cout, Q), dtype=torch.float32, device=rows.device)
        with torch.cuda.device(rows.device):
            rc = _lib.lib().rpe_pointconv_fused(
                _ptr(rows), CFp, M, _ptr(knn_indices), knn_indices.stride(1), _ptr(query_xyz), *query_xyz.stride(),
                _ptr(w["w1"]), _ptr(w["b1"]), _ptr(w["w2"]), _ptr(w["b2"]), 0.1, _ptr(w["lp"]), w["n_tiles"],
                _ptr(scale), _ptr(shift), _ACT[kind], 0.1, B, Q, Cout, int(out_rows), stride, _ptr(out), _lib.stream_of(rows))
        _lib.check(rc, "PointConv")
        return PackedRows(out, Cout) if out_rows else out


class PointConvDownSampling(_PointConv):
    """pointconv.py:7-61."""

    def forward(self, xyz, features, sampled_xyz, knn_indices=None, out_rows=False):
        """``features``: [B,C,M], a list of such tensors (concatenated), or PackedRows.  ``knn_indices`` [B,Q,>=k]: the
        neighbours of sampled_xyz in xyz if the caller already has them (the reference always searches here,
        pointconv.py:46).  ``out_rows``: return PackedRows over the SAMPLED points instead of [B,Cout,Q]."""
        if knn_indices is None:
            knn_indices = k_nearest_neighbor(xyz, sampled_xyz, self.k)  # [B,Q,k]
        return self._forward(xyz, features, sampled_xyz, knn_indices, out_rows)


class PointConvNoSampling(_PointConv):
    """pointconv.py:64-122."""

    def forward(self, xyz, features, knn_indices=None, out_rows=False):
        batch_size, n_points = xyz.shape[0], xyz.shape[2]
        if knn_indices is not None:
            assert knn_indices.shape[:2] == torch.Size([batch_size, n_points])
            assert knn_indices.shape[2] >= self.k
        else:
            knn_indices = k_nearest_neighbor(xyz, xyz, self.k)
        return self._forward(xyz, features, xyz, knn_indices, out_rows)
