"""RPEFlow model counterpart -- the CALLER of the hot path.

Same module tree, parameter names and shapes as the reference's ``models/RPEFlow.py`` +
``models/RPEFlow_core.py`` (so its checkpoints load with ``strict=True``), same forward
arithmetic at inference.  Every hot-path call -- FPS, KNN, correlation2d, warps, gathers,
PointConv, Correlation3D -- goes to the HIP kernels of this package; dense convolutions,
the Restormer cross-attention blocks and resizes stay on PyTorch-ROCm (MIOpen / hipBLASLt),
which SURVEY.md section 2 marks as outside the hot path.

Inference only.  The mutual-information heads (models/mutual_info.py) exist as parameter
holders so state dicts match, but are not evaluated: their output is a training loss that
never reaches the flows (RPEFlow_core.py:33-35, RPEFlow.py:95-99).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .csrc import correlation2d as native_correlation2d
from .csrc.wrapper import _correlation2d_algo as correlation2d_fused_leaky
from .hotpath import native_ops
from .pwc3d_core import FlowEstimator3D as NativeFlowEstimator3D
from .pwc3d_core import build_pc_pyramid as native_build_pc_pyramid
from .utils import Conv1dNormRelu, Conv2dNormRelu, conv_chain, conv_module, mesh_grid, resize_frames, run_chain, upsample2x_pair
from .utils import backwarp_2d as native_backwarp_2d
from .utils import grid_sample_sources, project_points
from .utils import grid_sample_wrapper as native_grid_sample_wrapper
from .utils import knn_interpolation as native_knn_interpolation
from .utils import project_feat_with_nn_corr as native_project_feat_with_nn_corr


class Config(dict):
    """Attribute-style nested dict (the reference uses omegaconf.DictConfig: cfgs.pwc2d.max_displacement ...)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = Config(v) if isinstance(v, dict) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None  # hasattr(), copy.deepcopy(), pickling rely on AttributeError


def things_config():
    """The ``model`` section of conf/test/things.yaml:13-45."""
    return Config({
        "name": "RPEFlow", "batch_size": 4, "freeze_bn": False,
        "ids": {"enabled": True, "sensor_size_divisor": 32},
        "attention": {"norm": "WithBias", "attention": "mdta", "layers": 2},
        "pwc2d": {"event_bins": 10, "event_polarity": True, "max_displacement": 4,
                  "norm": {"feature_pyramid": "batch_norm", "flow_estimator": None, "context_network": None}},
        "pwc3d": {"k": 16, "norm": {"feature_pyramid": "batch_norm", "correlation": None, "flow_estimator": None}},
    })


# ------------------------------------------------------------------ Restormer cross blocks (restormer_arch.py)
class _ChannelNorm(nn.Module):
    """LayerNorm over the channel axis of [B,C,...] (restormer_arch.py:31-83); parameters under ``body``."""

    class _Affine(nn.Module):
        def __init__(self, dim, with_bias):
            super().__init__()
            self.weight = nn.Parameter(torch.ones(dim))
            if with_bias:
                self.bias = nn.Parameter(torch.zeros(dim))

    def __init__(self, dim, kind):
        super().__init__()
        self.with_bias = kind != "BiasFree"
        self.body = self._Affine(dim, self.with_bias)

    def forward(self, x):
        if x.is_cuda:  # one kernel instead of mean / var / sub / sqrt / div / mul / add
            from .restormer_ops import channel_layernorm
            return channel_layernorm(x, self.body.weight, self.body.bias if self.with_bias else None)
        shape = [1, -1] + [1] * (x.dim() - 2)
        var = x.var(1, keepdim=True, unbiased=False)
        if self.with_bias:
            x = (x - x.mean(1, keepdim=True)) / torch.sqrt(var + 1e-5)
            return x * self.body.weight.view(shape) + self.body.bias.view(shape)
        return x / torch.sqrt(var + 1e-5) * self.body.weight.view(shape)


class _MutualAttention(nn.Module):
    """Mutual_Attention{2D,3D} (restormer_arch.py:169-204, 251-283): channel attention, q from x, k/v from y."""

    def __init__(self, dim, num_heads, bias, dims):
        super().__init__()
        conv = nn.Conv2d if dims == 2 else nn.Conv1d
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv_dwconv = conv(dim * 3, dim * 3, kernel_size=3, stride=1, padding=1, groups=dim * 3, bias=bias)
        self.project_out = conv(dim, dim, kernel_size=1, bias=bias)

    def forward(self, x, y, residual=None):
        """Returns project_out(attention) (+ residual when given)."""
        shape = x.shape
        if x.is_cuda and shape[1] // self.num_heads <= 96:
            # depth-wise conv reading x | y | y in place of the concatenation; gram + softmax + project_out folded into
            # one C x C matrix per sample (csrc/attention.hip); one batched GEMM applies it to v and adds the residual
            from .restormer_ops import attention_apply, channel_attention_matrix, dwconv3
            qkv = dwconv3([x, y, y], self.qkv_dwconv.weight, self.qkv_dwconv.bias)
            # the per-sample matrix in the 1x1 kernel's weight order: "residual + M[b] v[b] (+ bias)" is then ONE launch reading v
            # in place (was: a copy of the residual + a batched GEMM)
            m = channel_attention_matrix(qkv, self.num_heads, self.temperature, self.project_out.weight, packed=True)
            return attention_apply(qkv, m, residual=residual, bias=self.project_out.bias)
        out = self._forward_plain(x, y)
        return out if residual is None else residual + out

    def _forward_plain(self, x, y):
        shape = x.shape
        qkv = self.qkv_dwconv(torch.cat((x, y, y), dim=1))
        q, k, v = qkv.chunk(3, dim=1)
        heads = lambda t: t.reshape(shape[0], self.num_heads, shape[1] // self.num_heads, -1)
        q, k, v = F.normalize(heads(q), dim=-1), F.normalize(heads(k), dim=-1), heads(v)
        attn = ((q @ k.transpose(-2, -1)) * self.temperature).softmax(dim=-1)
        return self.project_out((attn @ v).reshape(shape))


class _GatedFeedForward(nn.Module):
    """FeedForward{2D,3D} (restormer_arch.py:88-110, 225-248)."""

    def __init__(self, dim, expansion, bias, dims):
        super().__init__()
        conv = nn.Conv2d if dims == 2 else nn.Conv1d
        hidden = int(dim * expansion)
        self.project_in = conv(dim, hidden * 2, kernel_size=1, bias=bias)
        self.dwconv = conv(hidden * 2, hidden * 2, kernel_size=3, stride=1, padding=1, groups=hidden * 2, bias=bias)
        self.project_out = conv(hidden, dim, kernel_size=1, bias=bias)

    def forward(self, x, residual=None, inplace=False):
        """project_out(gelu(a) * b) (+ residual, added by project_out's GEMM; ``inplace``: accumulated into ``residual``)."""
        if x.is_cuda:
            from .restormer_ops import dwconv3, gdfn_tail
            from .utils import _inference_only
            t = conv_module(self.project_in, x)
            if _inference_only(x, *self.parameters()):  # the point-cloud blocks: 3-tap conv + gelu gate + project_out (+ residual) in one launch
                out = gdfn_tail(t, self.dwconv.weight, self.dwconv.bias, self.project_out.weight, self.project_out.bias, residual, inplace)
                if out is not None:
                    return out
            hidden = dwconv3([t], self.dwconv.weight, self.dwconv.bias, gate=True)  # depth-wise conv + gelu gate in one kernel
            return conv_module(self.project_out, hidden, residual=residual, inplace=inplace)
        a, b = self.dwconv(self.project_in(x)).chunk(2, dim=1)
        out = self.project_out(F.gelu(a) * b)
        return out if residual is None else residual + out


class _CrossTransformerBlock(nn.Module):
    dims = 2

    def __init__(self, dim, num_heads, ffn_expansion_factor=2.66, bias=False, LayerNorm_type="WithBias"):
        super().__init__()
        self.norm1x = _ChannelNorm(dim, LayerNorm_type)
        self.norm1y = _ChannelNorm(dim, LayerNorm_type)
        self.attn = _MutualAttention(dim, num_heads, bias, self.dims)
        self.norm2 = _ChannelNorm(dim, LayerNorm_type)
        self.ffn = _GatedFeedForward(dim, ffn_expansion_factor, bias, self.dims)

    def forward(self, x, y):
        assert x.shape == y.shape
        if x.is_cuda:  # both LayerNorms in one launch
            from .restormer_ops import channel_layernorm_pair
            nx, ny = self.norm1x.body, self.norm1y.body
            nxo, nyo = channel_layernorm_pair(x, nx.weight, nx.bias if self.norm1x.with_bias else None,
                                              y, ny.weight, ny.bias if self.norm1y.with_bias else None)
        else:
            nxo, nyo = self.norm1x(x), self.norm1y(y)
        x = self.attn(nxo, nyo, residual=x)
        return self.ffn(self.norm2(x), residual=x, inplace=x.is_cuda)  # x is this block's own attention output: the GEMM adds into it


class CrossTransformerBlock2D(_CrossTransformerBlock):
    """restormer_arch.py:207-222."""
    dims = 2


class CrossTransformerBlock3D(_CrossTransformerBlock):
    """restormer_arch.py:287-302."""
    dims = 1


# ------------------------------------------------------------------ MI heads: parameters only (mutual_info.py)
class _MIHeads(nn.Module):
    def __init__(self, input_channels, hidden_channels, dims, event):
        super().__init__()
        block = Conv2dNormRelu if dims == 2 else Conv1dNormRelu
        for name in ["rgb", "point"] + (["event"] if event else []):
            setattr(self, name + "_mu", block(input_channels, hidden_channels, activation=None))
            setattr(self, name + "_logvar", block(input_channels, hidden_channels, activation=None))


def Mutual_info_reg_2D(c, h):
    return _MIHeads(c, h, 2, False)


def Mutual_info_reg_3D(c, h):
    return _MIHeads(c, h, 1, False)


def Mutual_info_reg_2D_Event(c, h):
    return _MIHeads(c, h, 2, True)


def Mutual_info_reg_3D_Event(c, h):
    return _MIHeads(c, h, 1, True)


# ------------------------------------------------------------------ 2-D blocks (pwc2d_core.py): plain convolutions
class ResidualBlock(nn.Module):
    """pwc2d_core.py:6-25 (down_sample=True is the only form the pyramid uses)."""

    def __init__(self, in_channels, out_channels, norm=None):
        super().__init__()
        self.down0 = Conv2dNormRelu(in_channels, out_channels, stride=2, norm=norm, activation=None)
        self.conv0 = Conv2dNormRelu(in_channels, out_channels, kernel_size=3, stride=2, padding=1, norm=norm)
        self.conv1 = Conv2dNormRelu(out_channels, out_channels, kernel_size=3, stride=1, padding=1, norm=norm, activation=None)
        self.relu = nn.LeakyReLU(negative_slope=0.1, inplace=True)

    def forward(self, x):
        from .utils import _inference_only, conv_no_bias_or
        epi1 = self.conv1._epilogue() if x.is_cuda and _inference_only(x, *self.parameters()) else None
        epi0 = self.down0._epilogue() if epi1 is not None else None
        if epi1 is None or epi0 is None or epi1[2] is not None or epi0[2] is not None:
            return self.relu(self.conv1(self.conv0(x)) + self.down0(x))
        # both branches' bias / BatchNorm, the sum and the activation in ONE pass over the two raw convolution outputs
        from .restormer_ops import channel_affine_add_act_, residual_tail_
        raw1 = conv_no_bias_or(self.conv1.conv_fn, self.conv0(x), False).contiguous()
        held = getattr(self, "_shift_of", (None, None))  # the two shift tensors the cached sum was made from (held, so
        if held[0] is not epi1[1] or held[1] is not epi0[1] or not hasattr(self, "_shift_sum"):  # their identity cannot be recycled)
            shifts = [t for t in (epi1[1], epi0[1]) if t is not None]
            self._shift_sum = (shifts[0] + shifts[1]) if len(shifts) == 2 else (shifts[0] if shifts else None)
            self._shift_of = (epi1[1], epi0[1])
        down = self.down0.conv_fn
        if (x.dtype == torch.float32 and x.is_contiguous() and down.kernel_size == (1, 1) and down.padding == (0, 0) and down.groups == 1
                and down.stride[0] == down.stride[1] and down.in_channels <= 256):
            # the strided 1x1 shortcut inside the tail's pass: no strided copy, no GEMM launch of its own
            return residual_tail_(raw1, epi1[0], self._shift_sum, x, down.weight, epi0[0], down.stride[0], "leaky_relu", 0.1)
        raw0 = conv_no_bias_or(down, x, False).contiguous()
        return channel_affine_add_act_(raw1, epi1[0], self._shift_sum, raw0, epi0[0], "leaky_relu", 0.1)


class FeaturePyramid2D(nn.Module):
    """pwc2d_core.py:28-40."""

    def __init__(self, n_channels, norm=None):
        super().__init__()
        self.pyramid_convs = nn.ModuleList(ResidualBlock(a, b, norm=norm) for a, b in zip(n_channels[:-1], n_channels[1:]))

    def forward(self, x):
        outputs = []
        for conv in self.pyramid_convs:
            x = conv(x)
            outputs.append(x)
        return outputs


class FlowEstimator2D(nn.Module):
    """pwc2d_core.py:92-135."""

    def __init__(self, n_channels, norm=None, conv_last=True):
        super().__init__()
        for i in range(1, 6):
            setattr(self, "conv%d" % i, Conv2dNormRelu(n_channels[i - 1], n_channels[i], kernel_size=3, padding=1, norm=norm))
        self.flow_feat_dim = n_channels[4] + n_channels[5]
        self.conv_last = nn.Conv2d(self.flow_feat_dim, 2, kernel_size=3, stride=1, padding=1) if conv_last else None

    def forward(self, x):
        x4 = conv_chain([self.conv1, self.conv2, self.conv3, self.conv4], x)
        flow_feat = torch.cat([self.conv5(x4), x4], dim=1)
        return (flow_feat, conv_module(self.conv_last, flow_feat)) if self.conv_last is not None else flow_feat


class ContextNetwork2D(nn.Module):
    """pwc2d_core.py:137-151."""

    def __init__(self, n_channels, dilations, norm=None):
        super().__init__()
        self.convs = nn.ModuleList(
            Conv2dNormRelu(a, b, kernel_size=3, padding=d, dilation=d, norm=norm)
            for a, b, d in zip(n_channels[:-1], n_channels[1:], dilations))
        self.conv_last = nn.Conv2d(n_channels[-1], 2, kernel_size=3, stride=1, padding=1)

    def forward(self, x, residual=None):
        """(features, conv_last(features) (+ residual: the flow the delta is added to, RPEFlow_core.py:419))."""
        x = conv_chain(self.convs, x)
        return x, conv_module(self.conv_last, x, residual=residual)


# ------------------------------------------------------------------ Bi-CLFM fusers (RPEFlow_core.py:14-162)
class PyramidFeatureFuser2D(nn.Module):
    def __init__(self, ops, in_channels_2d, in_channels_3d, num_heads, norm=None):
        super().__init__()
        self._ops = ops
        self.mlps = nn.Sequential(Conv2dNormRelu(in_channels_3d + 3, in_channels_2d, norm=norm))
        self.mi = Mutual_info_reg_2D(in_channels_2d, in_channels_2d // 2)
        self.fuse = CrossTransformerBlock2D(dim=in_channels_2d, num_heads=num_heads)

    def forward(self, xy, feat_2d, feat_3d, nn_proj, sampled_2d=None):
        """``sampled_2d``: grid_sample_wrapper(feat_2d, xy) when the caller has it (the 3-D fuser of the pair computes it)."""
        extra = {"sampled_2d": sampled_2d} if sampled_2d is not None else {}
        return self.fuse(feat_2d, self.mlps(self._ops.project_feat_with_nn_corr(xy, feat_2d, feat_3d, nn_proj[..., 0], **extra)))


class PyramidFeatureFuser3D(nn.Module):
    def __init__(self, ops, in_channels_2d, in_channels_3d, num_heads, norm=None):
        super().__init__()
        self._ops = ops
        self.mlps = nn.Sequential(Conv1dNormRelu(in_channels_2d, in_channels_3d, norm=norm))
        self.mi = Mutual_info_reg_3D(in_channels_3d, in_channels_3d // 2)
        self.fuse = CrossTransformerBlock3D(dim=in_channels_3d, num_heads=num_heads)

    def forward(self, xy, feat_2d, feat_3d, return_sampled=False):
        sampled = self._ops.grid_sample_wrapper(feat_2d, xy)
        fused = self.fuse(feat_3d, self.mlps(sampled))
        return (fused, sampled) if return_sampled else fused


class CorrFeatureFuser2D(nn.Module):
    def __init__(self, ops, in_channels_2d, in_channels_3d, num_heads):
        super().__init__()
        self._ops = ops
        self.mlps = nn.Sequential(
            Conv2dNormRelu(in_channels_3d * 2 + 5, in_channels_3d + in_channels_2d),
            Conv2dNormRelu(in_channels_3d + in_channels_2d, in_channels_2d))
        self.head_3d = Conv2dNormRelu(in_channels_3d + 5, in_channels_2d)
        self.head_event = Conv2dNormRelu(in_channels_3d, in_channels_2d)
        self.mi = Mutual_info_reg_2D_Event(in_channels_2d, in_channels_2d // 2)
        self.fuse = CrossTransformerBlock2D(dim=in_channels_2d, num_heads=num_heads)

    def forward(self, xy, feat_2d, feat_3d, efeat_2d, last_flow_2d, last_flow_3d_to_2d, nn_proj, flow_3d_scale=None):
        """``flow_3d_scale`` ((num_x, den_x), (num_y, den_y)): ``last_flow_3d_to_2d`` is the 3-D flow's xy still in sensor units; the
        conversion "* (image_w - 1) / (sensor_w - 1)" (RPEFlow_core.py:363-366) with the reference's two roundings as well as the
        cat with ``feat_3d`` (:373) happen as the projection kernel reads them."""
        project = self._ops.project_feat_with_nn_corr
        if feat_2d.is_cuda and project is native_project_feat_with_nn_corr:
            # also "-= last_flow_2d" on the projected flow (RPEFlow_core.py:82) and the cat with the event features (:83) inside the launch
            tail = {"feat_3d_tail": last_flow_3d_to_2d, "tail_scale": flow_3d_scale or (1.0, 1.0)}
            both = project(xy, feat_2d, feat_3d, nn_proj[..., 0], subtract_last=last_flow_2d, append=efeat_2d, **tail)
        else:
            if flow_3d_scale is not None:
                last_flow_3d_to_2d = _scaled_pair(last_flow_3d_to_2d, flow_3d_scale)
            feat_3d = torch.cat([feat_3d, last_flow_3d_to_2d], dim=1)
            feat_3d_to_2d = project(xy, feat_2d, feat_3d, nn_proj[..., 0])
            feat_3d_to_2d[:, -2:] -= last_flow_2d  # RPEFlow_core.py:82
            both = torch.cat([feat_3d_to_2d, efeat_2d], dim=1)
        return self.fuse(feat_2d, run_chain(self.mlps, both))


class CorrFeatureFuser3D(nn.Module):
    def __init__(self, ops, in_channels_2d, in_channels_3d, num_heads):
        super().__init__()
        self._ops = ops
        self.mlps = nn.Sequential(
            Conv1dNormRelu(in_channels_2d + in_channels_3d + 2, in_channels_2d + in_channels_3d),
            Conv1dNormRelu(in_channels_2d + in_channels_3d, in_channels_3d))
        self.head_2d = Conv1dNormRelu(in_channels_2d + 2, in_channels_3d)
        self.mi = Mutual_info_reg_3D_Event(in_channels_3d, in_channels_3d // 2)
        self.fuse = CrossTransformerBlock3D(dim=in_channels_3d, num_heads=num_heads)

    def forward(self, xy, feat_corr_2d, feat_corr_3d, efeat_2d, last_flow_3d, last_flow_2d_to_3d, flow_2d_scale=None):
        """``flow_2d_scale`` ((num_x, den_x), (num_y, den_y)): ``last_flow_2d_to_3d`` is the 2-D flow still in feature-map units; its
        conversion "* (sensor_w - 1) / (image_w - 1)" (RPEFlow_core.py:367-370, two roundings), both samplings, the subtraction (:110) and both concatenations (:105, :111) are ONE launch."""
        if feat_corr_2d.is_cuda and self._ops.grid_sample_wrapper is native_grid_sample_wrapper:
            both = grid_sample_sources([(feat_corr_2d, None, None), (last_flow_2d_to_3d, flow_2d_scale, last_flow_3d[:, :2]),
                                        (efeat_2d, None, None)], xy)
            return self.fuse(feat_corr_3d, run_chain(self.mlps, both))
        if flow_2d_scale is not None:
            last_flow_2d_to_3d = _scaled_pair(last_flow_2d_to_3d, flow_2d_scale)
        feat_2d_to_3d = self._ops.grid_sample_wrapper(torch.cat([feat_corr_2d, last_flow_2d_to_3d], dim=1), xy)
        efeat_2d_to_3d = self._ops.grid_sample_wrapper(efeat_2d, xy)
        feat_2d_to_3d[:, -2:] -= last_flow_3d[:, :2]  # RPEFlow_core.py:110
        return self.fuse(feat_corr_3d, run_chain(self.mlps, torch.cat([feat_2d_to_3d, efeat_2d_to_3d], dim=1)))


class DecoderFeatureFuser2D(nn.Module):
    def __init__(self, ops, in_channels_2d, in_channels_3d, num_heads):
        super().__init__()
        self._ops = ops
        self.mlps = nn.Sequential(Conv2dNormRelu(in_channels_3d + 3, in_channels_2d))
        self.mi = Mutual_info_reg_2D(in_channels_2d, in_channels_2d // 2)
        self.fuse = CrossTransformerBlock2D(dim=in_channels_2d, num_heads=num_heads)

    def forward(self, xy, feat_2d, feat_3d, nn_proj):
        return self.fuse(feat_2d, self.mlps(self._ops.project_feat_with_nn_corr(xy, feat_2d, feat_3d, nn_proj[..., 0])))


class DecoderFeatureFuser3D(nn.Module):
    def __init__(self, ops, in_channels_2d, in_channels_3d, num_heads):
        super().__init__()
        self._ops = ops
        self.mlps = nn.Sequential(Conv1dNormRelu(in_channels_2d, in_channels_3d))
        self.mi = Mutual_info_reg_3D(in_channels_3d, in_channels_3d // 2)
        self.fuse = CrossTransformerBlock3D(dim=in_channels_3d, num_heads=num_heads)

    def forward(self, xy, feat_2d, feat_3d):
        return self.fuse(feat_3d, self.mlps(self._ops.grid_sample_wrapper(feat_2d, xy)))


# ------------------------------------------------------------------ small geometry helpers (models/utils.py)
def project_pc2image(pc, camera_info):
    """utils.py:260-285."""
    if camera_info["projection_mode"] == "parallel":
        return torch.stack([pc[:, 0] + camera_info["cx"], pc[:, 1] + camera_info["cy"]], dim=1)
    f, cx, cy = (camera_info[k][:, None] for k in ("f", "cx", "cy"))
    return torch.stack([cx + (f / pc[:, 2]) * pc[:, 0], cy + (f / pc[:, 2]) * pc[:, 1]], dim=1)


def perspect2parallel(xyz, persp, paral):
    """IDS forward, utils.py:320-346."""
    f, cx, cy = (persp[k][:, None] for k in ("f", "cx", "cy"))
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    sw = (paral["sensor_w"] - 1) / (persp["sensor_w"] - 1)
    sh = (paral["sensor_h"] - 1) / (persp["sensor_h"] - 1)
    return torch.stack([
        (cx + (f / z) * x) * sw - (paral["sensor_w"] - 1) / 2,
        (cy + (f / z) * y) * sh - (paral["sensor_h"] - 1) / 2,
        (f * torch.log(z) + 1) * min(sw, sh)], dim=1)


def parallel2perspect(xyz, persp, paral):
    """IDS inverse, utils.py:349-377."""
    f, cx, cy = (persp[k][:, None] for k in ("f", "cx", "cy"))
    sw = (paral["sensor_w"] - 1) / (persp["sensor_w"] - 1)
    sh = (paral["sensor_h"] - 1) / (persp["sensor_h"] - 1)
    x = (xyz[:, 0] + (paral["sensor_w"] - 1) / 2) / sw
    y = (xyz[:, 1] + (paral["sensor_h"] - 1) / 2) / sh
    z = torch.exp((xyz[:, 2] / min(sw, sh) - 1) / f)
    return torch.stack([(x - cx) * z / f, (y - cy) * z / f, z], dim=1)


def resize_to_64x(x):
    """utils.py:227-241 (inputs only)."""
    h, w = x.shape[2:]
    if h % 64 == 0 and w % 64 == 0:
        return x
    return F.interpolate(x, size=((h + 63) // 64 * 64, (w + 63) // 64 * 64), mode="bilinear", align_corners=True)


def resize_flow2d(flow, target_h, target_w):
    """utils.py:217-224."""
    h, w = flow.shape[2:]
    if (h, w) == (target_h, target_w):
        return flow
    if flow.is_cuda:
        from .utils import resize_flow2d as fused
        return fused(flow, target_h, target_w)
    flow = F.interpolate(flow, size=(target_h, target_w), mode="bilinear", align_corners=True)
    flow *= _pair_scale(target_w / w, target_h / h, flow)
    return flow


def convex_upsample(flow, mask, scale_factor=8):
    """utils.py:201-214 (RAFT convex upsampling)."""
    if flow.is_cuda and scale_factor in (2, 4, 8):
        from .restormer_ops import convex_upsample as fused
        return fused(flow, mask, scale_factor)
    b, _, h, w = flow.shape
    mask = torch.softmax(mask.view(b, 1, 9, scale_factor, scale_factor, h, w), dim=2)
    up = F.unfold(flow * scale_factor, [3, 3], padding=1).view(b, 2, 9, 1, 1, h, w)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(b, 2, h * scale_factor, w * scale_factor)


# ------------------------------------------------------------------ the core (RPEFlow_core.py:165-432)
_pair_scale_cache = {}


def _scaled_pair(x, scale):
    """x[:, 0] * num_x / den_x, x[:, 1] * num_y / den_y for ``scale`` = ((num_x, den_x), (num_y, den_y)) -- the reference's
    "flow * (image_w - 1) / (sensor_w - 1)" (RPEFlow_core.py:363-370): a multiply, then a divide, each rounded; what the native
    kernels do as they read the flow (rpe_scaled)."""
    (nx, dx), (ny, dy) = scale
    return x * _pair_scale(nx, ny, x) / _pair_scale(dx, dy, x)


def _pair_scale(a, b, like):
    """[1,2,1(,1)] tensor (a, b) on like's device, cached: x[:, :2] * _pair_scale(...) scales the two flow channels by
    different factors in one launch.  (Created on the first, eager forward; graph capture then finds it resident.)"""
    key = (float(a), float(b), like.dim(), like.device)
    t = _pair_scale_cache.get(key)
    if t is None:
        t = _pair_scale_cache[key] = torch.tensor([a, b], dtype=torch.float32, device=like.device).view([1, 2] + [1] * (like.dim() - 2))
    return t


class StampTrace:
    """Timeline of a multi-stream forward: stamp(name) drops a one-thread kernel on the current stream that stores the
    GPU wall clock (rpe_clock_stamp).  Works inside a captured HIP graph: every replay refreshes the values.  Set
    rpeflow_amd.model.TRACE = StampTrace(device) before the (captured) forward, read() after a replay."""

    def __init__(self, device, slots=1024):
        self.buf = torch.zeros((slots, 2), dtype=torch.int64, device=device)  # per stamp: (engine cycles, constant-rate ticks)
        self.names = []

    def __call__(self, name):
        from . import _lib
        i = len(self.names)
        self.names.append(name)
        _lib.check(_lib.lib().rpe_clock_stamp(self.buf.data_ptr() + 16 * i, _lib.stream_of(self.buf)), "stamp")

    def read(self):
        """[(name, microseconds since the first stamp)] in issue order."""
        v = self.buf[:len(self.names), 1].cpu().tolist()  # (the constant-rate entries: ONE clock, unlike the cycle counters)
        return [(n, (t - v[0]) / 100.0) for n, t in zip(self.names, v)]


TRACE = None


def _stamp(name):
    if TRACE is not None:
        TRACE(name)


def _tensors(items):
    """The tensors among ``items`` (nested tuples flattened, None skipped): what crosses a stream boundary."""
    out = []
    for t in items:
        if torch.is_tensor(t):
            out.append(t)
        elif isinstance(t, (tuple, list)):
            out.extend(_tensors(t))
    return out


class _Branches:
    """Two-branch execution for decode(): at every pyramid level the 2-D chain (convolutions over H*W pixels) and the
    3-D chain (small kernels over N points) only meet at the three Bi-CLFM fusers, so the 3-D chain runs on a side HIP
    stream between those points.  Under graph capture the two chains become parallel branches of the one hipGraph.
    Tensors crossing streams are registered with the caching allocator (record_stream)."""

    def __init__(self, side_stream):
        self.side = side_stream
        self.main = torch.cuda.current_stream(side_stream.device) if side_stream is not None else None

    def fork(self, fn, inputs=()):
        """Run fn() on the side stream after everything queued on the main stream so far."""
        if self.side is None:
            return fn()
        self.side.wait_stream(self.main)
        for t in inputs:
            t.record_stream(self.side)
        with torch.cuda.stream(self.side):
            return fn()

    def join(self, outputs=()):
        """Main stream waits for the side branch; ``outputs`` are side-allocated tensors the main stream will read."""
        if self.side is not None:
            self.main.wait_stream(self.side)
            for t in outputs:
                t.record_stream(self.main)


class RPEFlow_core(nn.Module):
    def constants_intact(self):
        """True while every cached constant of decode() and of the point-cloud pyramid still holds the value it was made with
        (nobody wrote through a shared tensor).  Cheap enough for a test or a debugging assertion, one sync."""
        from . import pwc3d_core
        return all(int(z.count_nonzero()) == 0 for z in self._zeros.values()) and pwc3d_core.constants_intact()

    def clear_constants(self):
        """Drop the cached constants (they are rebuilt on the next forward).  Graphs captured before still hold theirs."""
        from . import pwc3d_core
        self._zeros.clear()
        pwc3d_core.clear_constants()

    def __init__(self, cfgs2d, cfgs3d, cfgsattention=None, ops=None):
        super().__init__()
        self.cfgs2d, self.cfgs3d = cfgs2d, cfgs3d
        self._native_ops = ops is None  # (a caller's own operators get fresh tensors where the native ones share constants)
        self.ops = ops = ops or native_ops()
        self._zeros = {}  # constant zero tensors of decode(), by (shape, dtype, device, inference mode)
        FeaturePyramid3D, Correlation3D, FlowEstimator3D = ops.FeaturePyramid3D, ops.Correlation3D, ops.FlowEstimator3D
        corr_ch = (2 * cfgs2d.max_displacement + 1) ** 2
        event_bins = cfgs2d.event_bins * 2 if cfgs2d.event_polarity else cfgs2d.event_bins
        widths = [32, 64, 96, 128, 192]

        def aligners(block):
            return nn.ModuleList([nn.Identity()] + [block(32, 64), block(64, 64), block(96, 64), block(128, 64), block(192, 64)])

        self.feature_pyramid_2d = FeaturePyramid2D([3, 16, 32, 64, 96, 128, 192], norm=cfgs2d.norm.feature_pyramid)
        self.feature_aligners_2d = aligners(Conv2dNormRelu)
        self.efeature_pyramid_2d = FeaturePyramid2D([event_bins, 32, 32, 64, 96, 128, 192], norm=cfgs2d.norm.feature_pyramid)
        self.efeature_aligners_2d = aligners(Conv2dNormRelu)
        self.flow_estimator_2d = FlowEstimator2D([64 + 64 + corr_ch + 2 + 32, 192, 128, 96, 64, 32],
                                                 norm=cfgs2d.norm.flow_estimator, conv_last=False)
        self.context_network_2d = ContextNetwork2D([self.flow_estimator_2d.flow_feat_dim + 2, 128, 128, 128, 96, 64, 32],
                                                   dilations=[1, 2, 4, 8, 16, 1], norm=cfgs2d.norm.context_network)
        self.up_mask_head_2d = nn.Sequential(nn.Conv2d(32, 256, kernel_size=3, stride=1, padding=1), nn.ReLU(inplace=True),
                                             nn.Conv2d(256, 4 * 4 * 9, kernel_size=1, stride=1, padding=0))

        self.feature_pyramid_3d = FeaturePyramid3D([16, 32, 64, 96, 128, 192], norm=cfgs3d.norm.feature_pyramid, k=cfgs3d.k)
        self.feature_aligners_3d = aligners(Conv1dNormRelu)
        self.correlations_3d = nn.ModuleList([nn.Identity()] + [Correlation3D(c, c, k=cfgs3d.k) for c in widths])
        self.correlation_aligners_3d = aligners(Conv1dNormRelu)
        self.flow_estimator_3d = FlowEstimator3D([64 + 64 + 3 + 64, 128, 128, 64], cfgs3d.norm.flow_estimator,
                                                 conv_last=False, k=cfgs3d.k)

        heads_p = [1, 2, 2, 4, 4]
        self.pyramid_feat_fusers_2d = nn.ModuleList([nn.Identity()] + [
            PyramidFeatureFuser2D(ops, c, c, num_heads=h, norm=cfgs2d.norm.feature_pyramid) for c, h in zip(widths, heads_p)])
        self.pyramid_feat_fusers_3d = nn.ModuleList([nn.Identity()] + [
            PyramidFeatureFuser3D(ops, c, c, num_heads=h, norm=cfgs3d.norm.feature_pyramid) for c, h in zip(widths, heads_p)])
        self.corr_feat_fusers_2d = nn.ModuleList([nn.Identity()] + [
            CorrFeatureFuser2D(ops, corr_ch, c, num_heads=h) for c, h in zip(widths, [1, 1, 3, 3, 3])])
        self.corr_feat_fusers_3d = nn.ModuleList([nn.Identity()] + [
            CorrFeatureFuser3D(ops, corr_ch, c, num_heads=h) for c, h in zip(widths, heads_p)])
        self.estimator_feat_fuser_2d = DecoderFeatureFuser2D(ops, self.flow_estimator_2d.flow_feat_dim, 64, num_heads=2)
        self.estimator_feat_fuser_3d = DecoderFeatureFuser3D(ops, self.flow_estimator_2d.flow_feat_dim, 64, num_heads=2)
        import inspect  # (a CPU port of the reference passed as ``ops`` has the reference's signature, without ``sampled_2d``)
        self._project_feat_params = tuple(inspect.signature(ops.project_feat_with_nn_corr).parameters)

        self.conv_last_2d = nn.Conv2d(self.flow_estimator_2d.flow_feat_dim, 2, kernel_size=3, stride=1, padding=1)
        self.conv_last_3d = nn.Conv1d(64, 3, kernel_size=1)

    def encode(self, image, xyzs):
        return self.feature_pyramid_2d(image), self.feature_pyramid_3d(xyzs)

    def encode_event(self, event_voxel):
        return self.efeature_pyramid_2d(event_voxel)

    def hoist(self, xyzs1, xyzs2, feats_2d_both, feats_3d_both, efeats_2d, camera_info, pre_stream=None):
        """Stage 1 of every level, hoisted out of the coarse-to-fine recurrence: projections, neighbour searches, pyramid
        fusers, aligners, Correlation3D's feature halves -- functions of the encoders' outputs only.  Issued on
        ``pre_stream`` when given (after everything queued on the current stream so far); decode() waits per level.
        ``efeats_2d`` None: the event pyramid is not there yet (forward() issues this call between the image and the event
        pyramid); decode() then aligns the event features itself."""
        sensor_h, sensor_w = camera_info["sensor_h"], camera_info["sensor_w"]
        k = self.cfgs3d.k
        k_nearest_neighbor = self.ops.k_nearest_neighbor
        camera_both = {key: (torch.cat([v, v]) if torch.is_tensor(v) else v) for key, v in camera_info.items()}
        top = len(xyzs1) - 1
        batch_size = feats_2d_both[1].shape[0] // 2

        def fuse_level(level):
            image_h, image_w = feats_2d_both[level].shape[2:]
            sx, sy = (image_w - 1) / (sensor_w - 1), (image_h - 1) / (sensor_h - 1)
            if xyzs1[level].is_cuda and self.ops.project_feat_with_nn_corr is native_project_feat_with_nn_corr:
                xy_both = project_points(xyzs1[level], xyzs2[level], camera_info, sx, sy)  # both frames, projection and rescale: one launch
            else:
                xy_both = project_pc2image(torch.cat([xyzs1[level], xyzs2[level]], dim=0), camera_both)
                xy_both *= _pair_scale(sx, sy, xy_both)
            grid = mesh_grid(2 * batch_size, image_h, image_w, xy_both.device).reshape(2 * batch_size, 2, -1)
            nn_proj_both = k_nearest_neighbor(xy_both, grid, k=1)
            knn_1in1 = k_nearest_neighbor(xyzs1[level], xyzs1[level], k=k)
            # (same map, same points, same stream: the 3-D fuser's samples serve the 2-D fuser's per-point rows)
            share = "sampled_2d" in self._project_feat_params
            fused_3d, sampled = self.pyramid_feat_fusers_3d[level](xy_both, feats_2d_both[level], feats_3d_both[level], return_sampled=True)
            fused_2d = self.pyramid_feat_fusers_2d[level](xy_both, feats_2d_both[level], feats_3d_both[level], nn_proj_both,
                                                          sampled_2d=sampled if share else None)
            # the aligners of the estimator inputs (:385-390) read the fused frame-1 features and the event features only
            aligned = (self.feature_aligners_2d[level](fused_2d[:batch_size]),
                       self.efeature_aligners_2d[level](efeats_2d[level]) if efeats_2d is not None else None,
                       self.feature_aligners_3d[level](fused_3d[:batch_size]))
            # the feat1 / feat2 halves of Correlation3D's first layer need the fused features only (pwc3d_core.Correlation3D)
            corr = self.correlations_3d[level]
            corr_proj = corr.project_stacked(fused_3d) if hasattr(corr, "project_stacked") else None
            return xy_both, nn_proj_both, knn_1in1, fused_2d, fused_3d, *aligned, corr_proj

        fused, ready = {}, {}
        if pre_stream is not None:
            main = torch.cuda.current_stream(pre_stream.device)
            pre_stream.wait_stream(main)
            for t in list(xyzs1) + list(xyzs2) + list(feats_2d_both) + list(feats_3d_both) + list(efeats_2d or []):
                t.record_stream(pre_stream)
            with torch.cuda.stream(pre_stream):
                for level in range(top, 0, -1):
                    fused[level] = fuse_level(level)
                    _stamp("pre L%d done" % level)
                    ready[level] = torch.cuda.Event()
                    ready[level].record(pre_stream)
        else:
            for level in range(top, 0, -1):
                fused[level] = fuse_level(level)
        return fused, ready

    def _up_mask(self, feat):
        """up_mask_head_2d (RPEFlow_core.py:208-211, 424): 3x3 convolution -> ReLU -> 1x1.  On the GPU the 3x3's bias and the ReLU are
        ONE in-place pass over the raw convolution output (the library adds the bias in a pass of its own and the ReLU is another:
        86 -> 45 us over [4,256,136,240] at the very end of the chain).  (Applied by the 1x1 kernel as it reads its input instead,
        the two passes disappear and that kernel goes from 147 to 201 us: its loop has no room for 64 more vector instructions.)"""
        head = self.up_mask_head_2d
        from .utils import _inference_only, conv_no_bias_or
        if feat.is_cuda and _inference_only(feat, *head.parameters()) and isinstance(head[1], nn.ReLU):
            return conv_module(head[2], conv_no_bias_or(head[0], feat, False, epilogue=(None, head[0].bias, "relu")))
        return conv_module(head[2], head[1](conv_module(head[0], feat)))

    def decode(self, xyzs1, xyzs2, feats_2d_both, feats_3d_both, efeats_2d, camera_info, side_stream=None, pre_stream=None,
               all_levels=False, hoisted_early=None):
        """RPEFlow_core.py:302-432 without the MI loss bookkeeping.

        ``feats_2d_both`` / ``feats_3d_both``: pyramids of frame 1 and frame 2 stacked on the batch axis ([2B,...], frame 1
        first).  The reference runs the shared-weight pyramid fusers once per frame (:329-338); all their layers are
        per-sample in eval mode, so one pass over the 2B stack computes the same thing with half the launches.
        ``side_stream``: run the 3-D chain of every level beside the 2-D chain (see _Branches); None = in order.
        ``pre_stream``: the pyramid fusers and neighbour searches of ALL levels depend on the encoders only, not on the
        coarse-to-fine flow recurrence; they are issued up front, on this third stream when given, and each level of the
        recurrence waits for its own set."""
        flows_2d, flows_3d, flow_feats_2d, flow_feats_3d = [], [], [], []
        sensor_h, sensor_w = camera_info["sensor_h"], camera_info["sensor_w"]
        md, k = self.cfgs2d.max_displacement, self.cfgs3d.k
        o = self.ops
        correlation2d, k_nearest_neighbor = o.correlation2d, o.k_nearest_neighbor
        backwarp_2d, backwarp_3d, knn_interpolation = o.backwarp_2d, o.backwarp_3d, o.knn_interpolation
        br = _Branches(side_stream)
        camera_both = {key: (torch.cat([v, v]) if torch.is_tensor(v) else v) for key, v in camera_info.items()}
        top = len(xyzs1) - 1
        batch_size = feats_2d_both[1].shape[0] // 2

        # ---- stage 1, hoisted out of the recurrence (hoist() above): issued here unless forward() did it earlier
        fused, ready = hoisted_early if hoisted_early is not None else self.hoist(xyzs1, xyzs2, feats_2d_both, feats_3d_both, efeats_2d, camera_info,
                                                                      pre_stream)

        def zeros(*shape):
            """The coarsest level's "previous" flows (RPEFlow_core.py:341-344): constant tensors, made once per shape -- every
            NATIVE consumer reads them (residual / subtrahend / gather source), none writes; ``constants_intact()`` checks it.
            Operators passed in by the caller (``ops=``) get a fresh tensor every time: nothing is known about what they write."""
            like = feats_2d_both[1]
            if not self._native_ops:
                return torch.zeros(shape, dtype=like.dtype, device=like.device)
            key = (shape, like.dtype, like.device, torch.is_inference_mode_enabled())  # (an inference tensor cannot serve a later autograd pass)
            if key not in self._zeros:
                z = torch.zeros(shape, dtype=like.dtype, device=like.device)
                if like.is_cuda and torch.cuda.is_current_stream_capturing():
                    return z  # (memory of the capturing graph's pool: not kept beyond this call)
                self._zeros[key] = z
            return self._zeros[key]
        main_stream = torch.cuda.current_stream(pre_stream.device) if pre_stream is not None else None

        def take(level):
            """This level's hoisted tensors, after the main stream has waited for them."""
            if pre_stream is not None:
                main_stream.wait_event(ready[level])
                for t in _tensors(fused[level]):
                    t.record_stream(main_stream)
            return fused.pop(level)

        def stage1_3d(level, hoisted):
            """3-D side of stage 1: up-sample the coarser 3-D flow, warp, cost volume (RPEFlow_core.py:345-361).  Reads the
            hoisted tensors and the side stream's own previous outputs only, so it follows the previous level's stage 3 on
            the side stream without waiting for the main stream's context network."""
            xy_both, _, knn_1in1, _, fused_3d = hoisted[:5]
            corr_proj = hoisted[8]
            xyz1, xyz2, n_points = xyzs1[level], xyzs2[level], xyzs1[level].shape[-1]
            image_h, image_w = feats_2d_both[level].shape[2:]
            sx, sy = (image_w - 1) / (sensor_w - 1), (image_h - 1) / (sensor_h - 1)
            if level == top:
                last_flow_3d, last_flow_feat_3d, xyz2_warp = zeros(batch_size, 3, n_points), zeros(batch_size, 64, n_points), xyz2
            else:
                _stamp("side L%d stage1 begin" % level)
                if xyz1.is_cuda and knn_interpolation is native_knn_interpolation:  # the pair is read where it lies: no cat
                    up = knn_interpolation(xyzs1[level + 1], (flows_3d[-1], flow_feats_3d[-1]), xyz1)
                else:
                    up = knn_interpolation(xyzs1[level + 1], torch.cat([flows_3d[-1], flow_feats_3d[-1]], dim=1), xyz1)
                _stamp("side L%d stage1 knn_interp done" % level)
                last_flow_3d, last_flow_feat_3d = up[:, :3, :], up[:, 3:, :]
                xyz2_warp = backwarp_3d(xyz1, xyz2, last_flow_3d)
                _stamp("side L%d stage1 backwarp done" % level)
            if corr_proj is not None:
                feat_corr_3d = self.correlations_3d[level](xyz1, fused_3d[:batch_size], xyz2_warp, fused_3d[batch_size:], knn_1in1,
                                                           projected=corr_proj)
            else:
                feat_corr_3d = self.correlations_3d[level](xyz1, fused_3d[:batch_size], xyz2_warp, fused_3d[batch_size:], knn_1in1)
            # the 2-D correlation fuser reads [cost volume | xy of the 3-D flow x (sx, sy)] (:371-373): product and cat happen
            # inside its projection kernel
            _stamp("side L%d stage1 done" % level)
            return last_flow_3d, last_flow_feat_3d, feat_corr_3d

        hoisted = take(top)
        out_s1 = br.fork(lambda: stage1_3d(top, hoisted), _tensors(hoisted) + [xyzs1[top], xyzs2[top]])
        for level in range(top, 0, -1):
            xyz1 = xyzs1[level]
            efeat_2d = efeats_2d[level]
            image_h, image_w = feats_2d_both[level].shape[2:]
            _stamp("main L%d start" % level)
            xy_both, nn_proj_both, knn_1in1, fused_2d, fused_3d, aligned_2d, aligned_e2d, aligned_3d, _ = hoisted
            if aligned_e2d is None:  # (hoisted before the event pyramid existed)
                aligned_e2d = self.efeature_aligners_2d[level](efeat_2d)
            xy1, nn_proj1 = xy_both[:batch_size], nn_proj_both[:batch_size]
            feat1_2d, feat2_2d_fused = fused_2d[:batch_size], fused_2d[batch_size:]

            # ---- stage 1, 2-D side: warp and cost volume
            if level == top:
                last_flow_2d, last_flow_feat_2d = zeros(batch_size, 2, image_h, image_w), zeros(batch_size, 32, image_h, image_w)
                feat2_2d_warp = feat2_2d_fused
            else:
                if flows_2d[-1].is_cuda and backwarp_2d is native_backwarp_2d:
                    last_flow_2d, last_flow_feat_2d = upsample2x_pair(flows_2d[-1], flow_feats_2d[-1], scale_a=2.0)
                else:
                    last_flow_2d = F.interpolate(flows_2d[-1] * 2, scale_factor=2, mode="bilinear", align_corners=True)
                    last_flow_feat_2d = F.interpolate(flow_feats_2d[-1], scale_factor=2, mode="bilinear", align_corners=True)
                feat2_2d_warp = backwarp_2d(feat2_2d_fused, last_flow_2d, padding_mode="border")
            if feat1_2d.is_cuda and correlation2d is native_correlation2d:  # leaky_relu of :362 fused into the kernel's epilogue
                feat_corr_2d = correlation2d_fused_leaky(feat1_2d, feat2_2d_warp, md, 0, leaky_slope=0.1)
            else:
                feat_corr_2d = F.leaky_relu(correlation2d(feat1_2d, feat2_2d_warp, md), 0.1)
            _stamp("main L%d stage1 done" % level)
            br.join(out_s1)
            last_flow_3d, last_flow_feat_3d, feat_corr_3d = out_s1
            sx, sy = (image_w - 1) / (sensor_w - 1), (image_h - 1) / (sensor_h - 1)

            # ---- stage 2: correlation fusers and flow estimators
            def chain_3d():
                # (the 2-D flow in sensor units, :103-104, exists only as the sampling kernel reads it)
                corr_3d_fused = self.corr_feat_fusers_3d[level](xy1, feat_corr_2d, feat_corr_3d, efeat_2d, last_flow_3d, last_flow_2d,
                                                                flow_2d_scale=((sensor_w - 1, image_w - 1), (sensor_h - 1, image_h - 1)))
                x_3d = [self.correlation_aligners_3d[level](corr_3d_fused), aligned_3d, last_flow_3d, last_flow_feat_3d]
                if not isinstance(self.flow_estimator_3d, NativeFlowEstimator3D):  # (the native one concatenates while packing)
                    x_3d = torch.cat(x_3d, dim=1)
                est = self.flow_estimator_3d(xyz1, x_3d, knn_1in1)
                _stamp("side L%d stage2 done" % level)
                return (est,)

            out_3d = br.fork(chain_3d, [feat_corr_2d, efeat_2d, last_flow_2d])
            corr_2d_fused = self.corr_feat_fusers_2d[level](xy1, feat_corr_2d, feat_corr_3d, efeat_2d, last_flow_2d, last_flow_3d[:, :2], nn_proj1,
                                                            flow_3d_scale=((image_w - 1, sensor_w - 1), (image_h - 1, sensor_h - 1)))
            x_2d = torch.cat([corr_2d_fused, aligned_2d, aligned_e2d, last_flow_2d, last_flow_feat_2d], dim=1)
            flow_feat_2d_raw = self.flow_estimator_2d(x_2d)
            _stamp("main L%d stage2 done" % level)
            br.join(out_3d)
            flow_feat_3d_raw, = out_3d

            # ---- stage 3: decoder fusers and flow heads.  The side stream goes straight on to stage 1 of the next level
            # (its inputs are hoisted or its own); the main stream meets it again at that level's cost volumes.
            nxt = take(level - 1) if level > 1 else None

            def chain_3d():
                flow_feat_3d = self.estimator_feat_fuser_3d(xy1, flow_feat_2d_raw, flow_feat_3d_raw)
                flow_3d = conv_module(self.conv_last_3d, flow_feat_3d, residual=last_flow_3d)
                _stamp("side L%d stage3 done" % level)
                flows_3d.append(flow_3d)
                flow_feats_3d.append(flow_feat_3d)
                if nxt is not None:
                    return stage1_3d(level - 1, nxt)
                # last level: the up-sampling of the 3-D flow to the full cloud belongs to this chain too (:430)
                return (knn_interpolation(xyzs1[1], flow_3d.float(), xyzs1[0]),)

            side_in = [flow_feat_2d_raw] + (_tensors(nxt) + [xyzs1[level - 1], xyzs2[level - 1]] if nxt is not None else [xyzs1[0]])
            out_s1 = br.fork(chain_3d, side_in)
            flow_feat_2d = self.estimator_feat_fuser_2d(xy1, flow_feat_2d_raw, flow_feat_3d_raw, nn_proj1)
            flow_2d = conv_module(self.conv_last_2d, flow_feat_2d, residual=last_flow_2d)
            flow_feat_2d, flow_2d = self.context_network_2d(torch.cat([flow_feat_2d, flow_2d], dim=1), residual=flow_2d)
            _stamp("main L%d stage3 done" % level)
            flows_2d.append(flow_2d)
            flow_feats_2d.append(flow_feat_2d)
            hoisted = nxt

        flows_2d = [f.float() for f in flows_2d][::-1]
        flows_2d[0] = convex_upsample(flows_2d[0], self._up_mask(flow_feats_2d[-1]), scale_factor=4)
        br.join(list(out_s1) + flows_3d + flow_feats_3d)
        flows_3d = [f.float() for f in flows_3d][::-1]
        flows_3d_up = [out_s1[0]]
        # the coarser levels' up-sampled flows (RPEFlow_core.py:426-430) feed the training losses only; inference reads [0]
        for i in range(1, len(flows_2d) if all_levels else 1):
            flows_2d[i] = F.interpolate(flows_2d[i] * 4, scale_factor=4, mode="bilinear", align_corners=True)
        for i in range(1, len(flows_3d) if all_levels else 1):
            flows_3d_up.append(knn_interpolation(xyzs1[i + 1], flows_3d[i], xyzs1[i]))
        return flows_2d, flows_3d_up + flows_3d[len(flows_3d_up):]


class RPEFlow(nn.Module):
    """models/RPEFlow.py:9-99, inference branch: forward(inputs) -> {'flow_2d', 'flow_3d'}.

    ``ids_on_host``: compute the IDS transform (log/div of 2*B*3*N floats) on the CPU, as the
    reference's CPU path does, so that FPS/KNN see bit-identical coordinates -- bit-identical to a
    reference run ON THE SAME HOST: torch.log differs by an ulp between CPU models.  An ``inputs["pcs_ids"]`` entry
    ([B,6,N], both clouds already transformed) bypasses the transform; the parity tests feed the reference's own
    coordinates that way."""

    def __init__(self, cfgs=None, ids_on_host=False, ops=None):
        """``ops``: namespace of hot-path callables/classes (rpeflow_amd.hotpath.OP_NAMES); default = this
        package's HIP-backed ones.  Tests and bench's cpu_baseline leg pass a CPU port instead."""
        super().__init__()
        self.cfgs = cfgs or things_config()
        self.ids_on_host = ids_on_host
        self.overlap_streams = True
        self.early_hoist = os.environ.get("RPE_EARLY_HOIST", "1") != "0"  # hoisted stage beside the event pyramid (forward())
        self.keep_levels = False  # also return every pyramid level's up-sampled flows, as decode() hands them over (tests)
        self._streams = {}
        self.pwc_fusion_core = RPEFlow_core(self.cfgs.pwc2d, self.cfgs.pwc3d, self.cfgs.get("attention"), ops=ops)

    def _pyramid(self, pc1, pc2, n_samples, fps_order):
        """(xyzs1, xyzs2, both): the two pyramids and, level by level, the two clouds stacked on the batch axis ([2B,3,n], frame 1
        first) as the shared-weight 3-D encoder takes them.  The native pyramid samples the stacked clouds with ONE gather, so
        every level of all three lists is a view of one tensor; a CPU port passed as ``ops`` gets the reference's calls."""
        build = self.pwc_fusion_core.ops.build_pc_pyramid
        extra = {} if fps_order is None else {"sample_index_both": fps_order}
        if build is native_build_pc_pyramid:
            xyzs1, xyzs2, _, _, both = build(pc1, pc2, n_samples, return_both=True, **extra)
            return xyzs1, xyzs2, both
        xyzs1, xyzs2, _, _ = build(pc1, pc2, n_samples, **extra)
        return xyzs1, xyzs2, [torch.cat([a, b], dim=0) for a, b in zip(xyzs1, xyzs2)]

    def _side_stream(self, device, name="side"):
        key = (torch.device(device).index, name)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    N_SAMPLES = [4096, 2048, 1024, 512, 256]

    def _cameras(self, inputs):
        """The perspective camera of the inputs and the parallel one of the IDS transform (RPEFlow.py:52-66); shapes only."""
        origin_h, origin_w = inputs["images"].shape[2:]
        intrinsics = inputs["intrinsics"]
        persp = {"projection_mode": "perspective", "sensor_h": origin_h, "sensor_w": origin_w,
                 "f": intrinsics[:, 0], "cx": intrinsics[:, 1], "cy": intrinsics[:, 2]}
        paral = None
        if self.cfgs.ids.enabled:
            div = self.cfgs.ids.sensor_size_divisor
            ph, pw = (origin_h + 63) // 64 * 64 // div, (origin_w + 63) // 64 * 64 // div
            paral = {"projection_mode": "parallel", "sensor_h": ph, "sensor_w": pw, "cx": (pw - 1) / 2, "cy": (ph - 1) / 2}
        return persp, paral

    def _clouds(self, inputs, persp, paral):
        """Both clouds as the pyramids see them: IDS-transformed when enabled (RPEFlow.py:60-66)."""
        device = inputs["images"].device
        pc1, pc2 = inputs["pcs"][:, :3], inputs["pcs"][:, 3:]
        if paral is None:
            return pc1, pc2
        if "pcs_ids" in inputs:  # the caller transformed the clouds already ([B,6,N]: frame 1, frame 2)
            return inputs["pcs_ids"][:, :3].to(device), inputs["pcs_ids"][:, 3:].to(device)
        if self.ids_on_host:
            host = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in persp.items()}
            return perspect2parallel(pc1.cpu(), host, paral).to(device), perspect2parallel(pc2.cpu(), host, paral).to(device)
        if pc1.is_cuda and self.pwc_fusion_core.ops.correlation2d is native_correlation2d:
            # one kernel, the reference's CPU rounding operation for operation (csrc/ids.hip): FPS is chaotic in these values
            from .utils import ids_forward
            both = ids_forward(inputs["pcs"], inputs["intrinsics"], persp, paral)
            return both[:pc1.shape[0]], both[pc1.shape[0]:]
        return perspect2parallel(pc1, persp, paral), perspect2parallel(pc2, persp, paral)

    @torch.no_grad()
    def sample_order(self, inputs):
        """[2B, 4096] int64: the furthest-point order of frame-1 then frame-2 clouds, exactly what forward() computes first
        (build_pc_pyramid, pwc3d_core.py:8-28).  It depends on ``pcs`` / ``intrinsics`` / the frame size only, so a caller
        that holds the NEXT batch can run it beside the current forward and hand the result over as
        ``inputs["fps_order"]`` (forward_ahead below)."""
        pc1, pc2 = self._clouds(inputs, *self._cameras(inputs))
        from .csrc import furthest_point_sampling
        return furthest_point_sampling(torch.cat([pc1, pc2], dim=0).transpose(1, 2), max(self.N_SAMPLES))

    @torch.no_grad()
    def forward_ahead(self, inputs, order, next_inputs):
        """forward(inputs) with the sampling software-pipelined by one batch.  ``order`` ([2B,4096] int64, caller-owned)
        holds sample_order(inputs) on entry and sample_order(next_inputs) on return: the 4095 dependent FPS iterations
        (3.5 ms on 2B of the 256 CUs, the longest serial stretch of the forward) run for the NEXT batch on their own
        stream under this batch's convolutions instead of in front of this batch's 3-D encoder.  Every call still runs
        one full FPS over one batch of clouds.  Capturable: the harness replays it as one HIP graph per batch."""
        mine = order.clone()
        main = torch.cuda.current_stream(order.device)
        ahead = self._side_stream(order.device, "ahead")
        ahead.wait_stream(main)
        with torch.cuda.stream(ahead):
            order.copy_(self.sample_order(next_inputs))
            _stamp("ahead sampling done")
        out = self.forward({**inputs, "fps_order": mine})
        main.wait_stream(ahead)
        return out

    @torch.no_grad()
    def forward(self, inputs, is_Train=False):
        raw, raw_events = inputs["images"], inputs["event_voxel"]
        origin_h, origin_w = raw.shape[2:]
        # frames 1 and 2 go through the shared-weight pyramids as one 2B batch (the reference calls encode() twice,
        # RPEFlow.py:78-79; eval-mode BatchNorm makes the two forms equal sample by sample)
        if raw.is_cuda and raw.dtype in (torch.uint8, torch.float32) and raw_events.dtype == torch.float32:
            # / 255, the resize to multiples of 64 and the frame split in one launch; the event grid in one more (or none)
            size = ((origin_h + 63) // 64 * 64, (origin_w + 63) // 64 * 64)
            image_both = resize_frames(raw, size, divisor=255.0, pair_split=True)
            event_voxel = raw_events if tuple(raw_events.shape[2:]) == size else resize_frames(raw_events, size)
        else:
            images = resize_to_64x(raw.float() / 255.0)
            event_voxel = resize_to_64x(raw_events)
            image_both = torch.cat([images[:, :3], images[:, 3:]], dim=0)
        persp, paral = self._cameras(inputs)
        pc1, pc2 = self._clouds(inputs, persp, paral)
        fps_order = inputs.get("fps_order")

        core = self.pwc_fusion_core
        n_samples = self.N_SAMPLES
        _stamp("main start")

        camera = paral if self.cfgs.ids.enabled else persp
        early = None
        if pc1.is_cuda and self.overlap_streams:
            # The 3-D encoder (FPS: 4096 dependent samples on 2B workgroups, then small PointConv kernels) and
            # the 2-D pyramids (large convolutions) share no data until decode(): run them on two HIP
            # streams.  FPS alone keeps 2B of 256 CUs busy for ~3.5 ms; here it hides behind the convolutions.
            main = torch.cuda.current_stream(pc1.device)
            side = self._side_stream(pc1.device)
            pre = self._side_stream(pc1.device, "pre")
            side.wait_stream(main)
            with torch.cuda.stream(side):
                xyzs1, xyzs2, both = self._pyramid(pc1, pc2, n_samples, fps_order)
                _stamp("side fps done")
                if self.early_hoist:
                    feats_3d_both = core.feature_pyramid_3d(both)
                    _stamp("side encode3d done")
                    encoded_3d = torch.cuda.Event()
                    encoded_3d.record(side)
            feats_2d_both = core.feature_pyramid_2d(image_both)
            _stamp("main image pyramid done")
            if self.early_hoist:
                # The hoisted stage of every level reads the image and point pyramids only: it starts here, beside the event
                # pyramid, instead of behind it (the coarsest level's set was what the first decoder level waited for).
                main.wait_event(encoded_3d)
                for t in list(xyzs1) + list(xyzs2) + list(feats_3d_both):
                    t.record_stream(main)  # allocated on the side stream, consumed on the main one
                early = core.hoist(xyzs1, xyzs2, feats_2d_both, feats_3d_both, None, camera, pre_stream=pre)
            efeats_2d = core.encode_event(event_voxel)
            _stamp("main event pyramid done")
            if not self.early_hoist:
                with torch.cuda.stream(side):
                    feats_3d_both = core.feature_pyramid_3d(both)
                    _stamp("side encode3d done")
                main.wait_stream(side)
                for t in list(xyzs1) + list(xyzs2) + list(feats_3d_both):
                    t.record_stream(main)  # allocated on the side stream, consumed on the main one
        else:
            xyzs1, xyzs2, both = self._pyramid(pc1, pc2, n_samples, fps_order)
            feats_3d_both = core.feature_pyramid_3d(both)
            _stamp("side encode3d done")
            feats_2d_both = core.feature_pyramid_2d(image_both)
            _stamp("main image pyramid done")
            efeats_2d = core.encode_event(event_voxel)
            _stamp("main event pyramid done")
        side = self._side_stream(pc1.device) if (pc1.is_cuda and self.overlap_streams) else None
        pre = self._side_stream(pc1.device, "pre") if side is not None else None
        flows_2d, flows_3d = core.decode(xyzs1, xyzs2, feats_2d_both, feats_3d_both, efeats_2d, camera, side_stream=side, pre_stream=pre,
                                         all_levels=self.keep_levels, hoisted_early=early)
        _stamp("main decode done")
        flow_3d = flows_3d[0]
        if self.cfgs.ids.enabled:
            xyz1 = xyzs1[0]
            if xyz1.is_cuda and core.ops.correlation2d is native_correlation2d:
                from .utils import ids_flow_inverse
                flow_3d = ids_flow_inverse(xyz1, flow_3d, inputs["intrinsics"], persp, paral)
            else:
                flow_3d = parallel2perspect(xyz1 + flow_3d, persp, paral) - parallel2perspect(xyz1, persp, paral)
        out = {"flow_2d": resize_flow2d(flows_2d[0], origin_h, origin_w), "flow_3d": flow_3d}
        if self.keep_levels:  # what RPEFlow_core.decode returns (RPEFlow_core.py:432), 3-D flows still in IDS space
            out["levels_2d"], out["levels_3d"] = list(flows_2d), list(flows_3d)
        return out
