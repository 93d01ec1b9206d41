"""Host-side mirror of the hot-path functions of the reference's models/utils.py.

Same names and argument meaning as the reference (file:line cited per function);
each call is ONE HIP kernel from librpeflow_hip.so (plus the KNN where the
reference also runs one).  GPU tensors only; see rpeflow_amd/_lib.py.

The 1x1-conv helpers (Conv1dNormRelu, Conv2dNormRelu, MLP1d, MLP2d; utils.py:7-98)
are dense convolutions and stay on PyTorch (MIOpen/hipBLASLt); they are restated
here only so that module trees and state-dict keys match the reference.
"""
import ctypes

import os

import torch
import torch.nn as nn

from . import _lib
from .csrc import k_nearest_neighbor

_NULL = ctypes.c_void_p(0)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


def _launch(dev_tensor, name, fn, *args):
    with torch.cuda.device(dev_tensor.device):
        rc = fn(*args, _lib.stream_of(dev_tensor))
    _lib.check(rc, name)


# ------------------------------------------------------------------ 1x1 conv stacks (PyTorch)
def _norm(kind, channels, dims):
    if kind == "batch_norm":
        return (nn.BatchNorm1d if dims == 1 else nn.BatchNorm2d)(channels)
    if kind == "instance_norm":
        return (nn.InstanceNorm1d if dims == 1 else nn.InstanceNorm2d)(channels)
    if kind is None:
        return nn.Identity()
    raise NotImplementedError("Unknown normalization function: %s" % kind)


def _act(kind):
    if kind == "relu":
        return nn.ReLU(inplace=True)
    if kind == "leaky_relu":
        return nn.LeakyReLU(negative_slope=0.1, inplace=True)
    if kind is None:
        return nn.Identity()
    raise NotImplementedError("Unknown activation function: %s" % kind)


def _inference_only(*tensors):
    """The fused paths work on detached parameter copies and build no autograd graph: they may only run when nothing
    would be differentiated -- grad mode off, or neither the input nor a parameter requires grad."""
    return not torch.is_grad_enabled() or not any(t is not None and t.requires_grad for t in tensors)


# flops per launch up to which the one-launch kernel is used.  Measured on the forward (median step of 40, two runs each): never
# 17.44 ms, <= 0.6e9 17.60, <= 2.5e9 17.33, <= 1e10 17.32, always 17.21 -- even the level-1 projections (13.5 GFLOP), where the
# library GEMM alone is faster, gain more from the epilogue / residual pass they no longer need.  The library path stays for
# anything larger than the model produces.
_PW_MAX_FLOPS = float(os.environ.get("RPE_POINTWISE_MAX_FLOPS", 1e12))


def _pw_packed_weight(weight):
    """The 1x1 weight [Cout,Cin,...] in rpe_pointwise_conv's fragment order (include/rpeflow_hip.h), cached ON the tensor object
    until it changes (a cache keyed by address would outlive the tensor: the allocator hands the same block to the next one)."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape), weight.device)
    hit = getattr(weight, "_rpe_pw_packed", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    cout = weight.shape[0]
    cin = weight.numel() // cout  # (a k x k weight counts as Cin*k*k input channels: the rows of its im2col matrix)
    nt, kt = (cout + 15) // 16, (cin + 3) // 4
    w = torch.zeros((16 * nt, 4 * kt), dtype=torch.float32, device=weight.device)
    w[:cout, :cin] = weight.detach().reshape(cout, cin).float()
    packed = w.reshape(nt, 16, kt, 4).permute(0, 2, 3, 1).contiguous()  # [ot][kt][k][i]
    weight._rpe_pw_packed = (key, packed)
    return packed


def pointwise_conv(x, weight, bias=None, stride=1, residual=None, inplace=False, epilogue=None):
    """A 1x1 convolution (Conv1d / Conv2d, groups 1, no padding) over [B,C,P] (stride s reads every s-th pixel first), with
    ``epilogue`` = (scale, shift, act) of the ConvNormRelu block around it and an optional ``residual`` [B,Cout,...].

    Layers up to _PW_MAX_FLOPS (every one of this model's) run in ONE launch of rpe_pointwise_conv -- GEMM, bias / BatchNorm,
    activation and residual add together (csrc/pointwise.hip).  Larger ones are ONE rocBLAS strided-batched GEMM
    y[b] = W @ x[b] (the residual accumulated by the GEMM, beta = 1), followed by the epilogue pass if there is one.

    Why not the convolution call: MIOpen's own choice for most of this model's ~110 1x1 shapes is that very GEMM
    (GemmFwd1x1_0_1), but which solver a process gets is decided by a timing race when it first meets a shape, and some of
    the candidates (GemmFwd1x1_0_2 on the small stride-2 layers, the splitting implicit GEMMs) accumulate with atomics:
    see the note above wants_im2col.  Both paths here have a fixed summation order and no timing-based selection: same
    bits in every process and every replay."""
    if stride != 1:
        x = x[(slice(None), slice(None)) + (slice(None, None, stride),) * (x.dim() - 2)]
    B, C = x.shape[0], x.shape[1]
    spatial = x.shape[2:]
    xf = x.reshape(B, C, -1)  # (a strided view is copied here: the subsampled pixels, nothing else)
    cout, P = weight.shape[0], xf.shape[2]
    scale, shift, kind = epilogue if epilogue is not None else (None, None, None)
    if bias is not None:  # (an epilogue's shift already contains the block's bias: callers pass one or the other)
        shift = bias if shift is None else shift + (bias if scale is None else bias * scale)
    if x.is_cuda and 2.0 * B * P * C * cout <= _PW_MAX_FLOPS:
        # a channel slice of a wider tensor (samples dense, batch stride larger) is read / added where it lies.  PyTorch's
        # stride of a size-1 dimension is arbitrary (B == 1, one channel, one position): such a dimension is not looked at,
        # and the batch stride handed to the kernel is the dense one then
        dense = lambda t, c: (P == 1 or t.stride(2) == 1) and (c == 1 or t.stride(1) == P) and (B == 1 or t.stride(0) >= c * P)
        batch_stride = lambda t, c: t.stride(0) if B > 1 else c * P
        xf = _f32(xf)
        if not dense(xf, C):
            xf = xf.contiguous()
        res = None
        if residual is not None:
            res = _f32(residual).reshape(B, cout, P)
            if not dense(res, cout):
                res = res.contiguous()
        reuse = inplace and res is not None and res.is_contiguous() and res.data_ptr() == residual.data_ptr()
        out = res if reuse else torch.empty((B, cout, P), dtype=torch.float32, device=x.device)
        _launch(xf, "pointwise_conv", _lib.lib().rpe_pointwise_conv, _ptr(xf), batch_stride(xf, C), B, C, P, _ptr(_pw_packed_weight(weight)), 0, cout,
                _ptr(scale) if scale is not None else _NULL, _ptr(shift.contiguous()) if shift is not None else _NULL,
                _ACT_CODE[kind], 0.1, _ptr(res) if res is not None else _NULL, batch_stride(res, cout) if res is not None else 0, _ptr(out))
        return out.reshape((B, cout) + tuple(spatial))
    w = weight.reshape(cout, -1)
    wb = w.unsqueeze(0).expand(B, -1, -1)  # batch stride 0: one strided-batched GEMM, the weight read once per sample from L2
    # (torch.matmul(w, xf) would fold the batch into the columns instead: a transposed copy of x in, one of y out)
    add = None if residual is None else residual.reshape(B, cout, -1)
    plain_shift = scale is None and kind is None  # nothing but an additive term: the GEMM can carry it
    if shift is not None and plain_shift:
        add = shift.view(1, -1, 1) if add is None else add + shift.view(1, -1, 1)
    with _rocblas():
        if add is None:
            y = torch.bmm(wb, xf)
        elif inplace and residual is not None and (shift is None or not plain_shift) and residual.is_contiguous():
            y = add.baddbmm_(wb, xf)  # accumulates into ``residual`` itself: no copy of it in front of the GEMM (the caller owns it)
        else:
            y = torch.baddbmm(add, wb, xf)
    if not plain_shift:
        assert residual is None, "pointwise_conv: epilogue + residual on the library path is not used by the model"
        from .restormer_ops import channel_affine_act_
        y = channel_affine_act_(y.contiguous(), scale, shift, kind, 0.1)
    return y.reshape((B, cout) + tuple(spatial))


class _rocblas:
    """These GEMMs through rocBLAS, as MIOpen's own GemmFwd1x1 solver runs them, not hipBLASLt (PyTorch's default on this
    GPU): fp32 rocBLAS kernels are built on v_mfma_f32_16x16x4 and measure 1.0-1.4x faster on the model's shapes, 16x on
    the 510 -> 192 projections over 135 positions (hipBLASLt: 16x16x1 tiles; tools/experiments/blas_pref.py).  Host-side dispatch
    state only: safe under graph capture."""
    lib = os.environ.get("RPE_POINTWISE_BLAS", "cublas")  # PyTorch's name for rocBLAS; "cublaslt": leave the default

    def __enter__(self):
        self.prev = None
        if self.lib != "cublaslt":
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self.prev = torch.backends.cuda.preferred_blas_library()
                torch.backends.cuda.preferred_blas_library(self.lib)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.backends.cuda.preferred_blas_library(self.prev)


def is_pointwise(conv):
    """A Conv1d / Conv2d module that goes through pointwise_conv whatever its input size: 1x1 kernel, stride 1, one group, no
    padding.  (Stride-2 1x1 convolutions do so on small maps only, conv_no_bias_or: on the large ones MIOpen's Winograd
    kernel beats gather + GEMM -- 42 vs 66 us at 288 x 480 -- and is deterministic.)"""
    return (all(k == 1 for k in conv.kernel_size) and conv.groups == 1 and isinstance(conv.padding, tuple)
            and all(p == 0 for p in conv.padding) and all(st == 1 for st in conv.stride))


# MIOpen runs SMALL convolutions it cannot give to Winograd (dilated, strided; sometimes plain 3x3 ones on a 9 x 15 map) as
# implicit GEMMs that split the reduction over workgroups and add the partial sums with fp32 ATOMICS (igemm ... _gkgs,
# GemmFwd1x1_0_2): the summation order, and with it the last bits of the result, changes from launch to launch.  Fifteen of
# the forward's 194 convolution shapes did (tools/experiments/conv_determinism.py) -- the context network's dilated 3x3 layers and
# the pyramids' last stride-2 layers, up to 72 x 120 -- and the decoder amplifies those last bits: 40 replays of ONE graph on
# ONE batch gave 35 different outputs, max |d flow_2d| 4e-4 ... 2.7e-2, |dEPE2D| against the reference 2e-6 ... 7.6e-5 (what
# rounds 1-2 read as two "solver populations").  torch.backends.cudnn.deterministic makes MIOpen avoid those kernels at 3x
# their time (10x for the whole forward).  Here such a convolution is im2col (F.unfold) + one rocBLAS strided-batched GEMM:
# fixed summation order, +10 us on the smallest maps, equal from 36 x 60 on (+0.13 ms per forward in all).
_IM2COL_MAX_POSITIONS = 36000   # batch x output positions: beyond, MIOpen's implicit GEMMs fill the GPU without splitting K
_IM2COL_TINY_POSITIONS = int(os.environ.get("RPE_IM2COL_TINY", 2200))  # plain 3x3 convolutions up to 18 x 30, batch 4: sometimes given to the splitting kernels too, and (with the epilogue handed to the next unfold, conv_chain) faster than Winograd + its epilogue pass on maps this small: 15.92 -> 15.83 ms per batch; 8700 (36 x 60): no gain
_IM2COL_MAX_BYTES = 192 << 20   # size of the unfolded input
# The unfolded GEMM stays on rocBLAS + one epilogue pass: K = 9 Cin is deep (1152 for the context network), where the 1x1 kernel
# (no K split, four channel groups in flight) loses: forward 17.2 ms with the library GEMM, 18.3 fused below 1 GFLOP, 18.6 always.
_IM2COL_FUSED_MAX_FLOPS = float(os.environ.get("RPE_IM2COL_FUSED_MAX_FLOPS", 0))


def _out_size(n, k, s, p, d):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


def wants_im2col(conv, x):
    """A Conv2d call MIOpen would (or might) run with atomic split-K accumulation: see above."""
    if x.dim() != 4 or conv.groups != 1 or all(k == 1 for k in conv.kernel_size) or not isinstance(conv.padding, tuple):
        return False
    ho = _out_size(x.shape[2], conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0])
    wo = _out_size(x.shape[3], conv.kernel_size[1], conv.stride[1], conv.padding[1], conv.dilation[1])
    positions = x.shape[0] * ho * wo
    special = any(d > 1 for d in conv.dilation) or any(st > 1 for st in conv.stride)
    cols = positions * x.shape[1] * conv.kernel_size[0] * conv.kernel_size[1] * 4
    return positions <= _IM2COL_MAX_POSITIONS and cols <= _IM2COL_MAX_BYTES and (special or positions <= _IM2COL_TINY_POSITIONS)


def im2col_conv(x, weight, bias, stride, padding, dilation, epilogue=None, pending=None):
    """conv2d as im2col + ONE GEMM with a fixed summation order: rpe_im2col, then one rocBLAS strided-batched GEMM over the
    unfolded input and the block's epilogue pass (the 1x1 kernel with the epilogue inside is measurably slower at K = 9 Cin).
    ``pending``: the (scale, shift, act) epilogue the layer that produced ``x`` still owes; the unfold applies it to what it reads."""
    B, _, H, W = x.shape
    k = weight.shape
    ho = _out_size(H, k[2], stride[0], padding[0], dilation[0])
    wo = _out_size(W, k[3], stride[1], padding[1], dilation[1])
    if x.is_cuda:  # [B, C*kh*kw, ho*wo] in one launch (F.unfold: one per sample)
        x = _f32(x).contiguous()  # the kernel indexes a dense [B, C, H, W]: channels_last or channel-sliced inputs are copied once
        cols = torch.empty((B, k[1] * k[2] * k[3], ho * wo), dtype=torch.float32, device=x.device)
        scale, shift, act = pending if pending is not None else (None, None, None)
        _launch(x, "im2col", _lib.lib().rpe_im2col, _ptr(x), B, k[1], H, W, k[2], k[3], stride[0], stride[1], padding[0], padding[1],
                dilation[0], dilation[1], _ptr(scale), _ptr(shift), {None: 0, "relu": 1, "leaky_relu": 2}[act], 0.1, _ptr(cols))
    else:
        assert pending is None
        cols = torch.nn.functional.unfold(x, (k[2], k[3]), dilation=dilation, padding=padding, stride=stride)
    if x.is_cuda and 2.0 * cols.numel() * k[0] <= _IM2COL_FUSED_MAX_FLOPS:
        return pointwise_conv(cols, weight, bias, 1, epilogue=epilogue).reshape(B, k[0], ho, wo)
    wb = weight.reshape(1, k[0], -1).expand(B, -1, -1)
    with _rocblas():
        y = torch.bmm(wb, cols) if bias is None else torch.baddbmm(bias.view(1, -1, 1), wb, cols)
    return _apply_epilogue(y.reshape(B, k[0], ho, wo), epilogue)


def conv_no_bias_or(conv, x, with_bias, epilogue=None):
    """The convolution of an nn.Conv1d / nn.Conv2d on the GPU, outside autograd: 1x1 -> pointwise_conv, small dilated / strided
    / tiny -> im2col_conv (both deterministic GEMMs), everything else MIOpen.  ``epilogue`` (scale, shift, act): applied by the
    call (fused into the 1x1 kernel where that runs, one in-place pass otherwise)."""
    bias = conv.bias if with_bias else None
    y = _conv_paths(conv, x, bias, epilogue)
    return y


def _apply_epilogue(y, epilogue):
    if epilogue is None or epilogue == (None, None, None):
        return y
    from .restormer_ops import channel_affine_act_
    return channel_affine_act_(y.contiguous(), epilogue[0], epilogue[1], epilogue[2], 0.1)


def _conv_paths(conv, x, bias, epilogue):
    if is_pointwise(conv):
        return pointwise_conv(x, conv.weight, bias, 1, epilogue=epilogue)
    if (all(k == 1 for k in conv.kernel_size) and conv.groups == 1 and x.dim() == 4
                              and x.shape[0] * x.shape[2] * x.shape[3] <= 4 * _IM2COL_MAX_POSITIONS and isinstance(conv.padding, tuple)
                              and all(p == 0 for p in conv.padding) and len(set(conv.stride)) == 1):
        return pointwise_conv(x, conv.weight, bias, conv.stride[0], epilogue=epilogue)  # (the small stride-2 1x1 layers)
    if wants_im2col(conv, x):
        return im2col_conv(x, conv.weight, bias, conv.stride, conv.padding, conv.dilation, epilogue)
    f = torch.nn.functional.conv1d if x.dim() == 3 else torch.nn.functional.conv2d
    return _apply_epilogue(f(x, conv.weight, bias, conv.stride, conv.padding, conv.dilation, conv.groups), epilogue)


def conv_module(conv, x, residual=None, inplace=False):
    """conv(x) (+ residual) for an nn.Conv1d / nn.Conv2d; on the GPU outside autograd through the deterministic paths above,
    a 1x1 convolution adding the residual inside its GEMM (``inplace``: into the residual tensor itself)."""
    if x.is_cuda and _inference_only(x, *conv.parameters()):
        if residual is not None and is_pointwise(conv):
            return pointwise_conv(x, conv.weight, conv.bias, 1, residual=residual, inplace=inplace)
        if residual is not None and residual.shape[1] == conv.out_channels and residual.dtype == torch.float32:
            # bias and residual in ONE pass over the raw convolution output (was: the library's bias kernel + an add)
            from .restormer_ops import channel_affine_add_act_
            y = conv_no_bias_or(conv, x, False).contiguous()
            if y.shape == residual.shape:
                return channel_affine_add_act_(y, None, conv.bias, residual.contiguous(), None, None)
            return residual + (y if conv.bias is None else y + conv.bias.view((1, -1) + (1,) * (y.dim() - 2)))
        y = conv_no_bias_or(conv, x, True)
    else:
        y = conv(x)
    return y if residual is None else residual + y


def affine_epilogue(owner, bias, norm, act):
    """(scale, shift, act) of the per-channel epilogue y = act(scale*x + shift) that a bias add, eval-mode BatchNorm
    and the activation amount to -- or None when that does not apply (training BN, InstanceNorm).  Cached on ``owner``
    until a parameter or buffer changes."""
    if isinstance(norm, nn.modules.batchnorm._BatchNorm):
        if norm.training or not norm.track_running_stats:
            return None
    elif not isinstance(norm, nn.Identity):
        return None
    kind = "leaky_relu" if isinstance(act, nn.LeakyReLU) else "relu" if isinstance(act, nn.ReLU) else None
    if kind is None and not isinstance(act, nn.Identity):
        return None
    tensors = [t for t in (bias, *norm.parameters(), *norm.buffers()) if t is not None]
    key = tuple((t.data_ptr(), t._version) for t in tensors)
    cache = getattr(owner, "_epi_cache", None)
    if cache is None or cache[0] != key:
        bias = bias.detach().float() if bias is not None else None
        scale = None
        if isinstance(norm, nn.modules.batchnorm._BatchNorm):
            scale = (norm.weight.detach() if norm.affine else 1.0) / torch.sqrt(norm.running_var + norm.eps)
            shift = (norm.bias.detach() if norm.affine else 0.0) - norm.running_mean * scale
            if bias is not None:
                shift = shift + bias * scale
            scale, bias = scale.float().contiguous(), shift.float().contiguous()
        owner._epi_cache = cache = (key, (scale, bias.contiguous() if bias is not None else None, kind))
    return cache[1]


class _ConvNormRelu(nn.Module):
    """utils.py:7-62: conv_fn / norm_fn / relu_fn, in that order and under those names."""
    dims = 1

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, groups=1,
                 norm=None, activation="leaky_relu"):
        super().__init__()
        conv = nn.Conv1d if self.dims == 1 else nn.Conv2d
        self.conv_fn = conv(in_channels, out_channels, kernel_size, stride, padding, dilation, groups)
        self.norm_fn = _norm(norm, out_channels, self.dims)
        self.relu_fn = _act(activation)

    def _epilogue(self):
        return affine_epilogue(self, self.conv_fn.bias, self.norm_fn, self.relu_fn)

    def forward(self, x):
        epi = self._epilogue() if x.is_cuda and _inference_only(x, *self.parameters()) else None
        if epi is None:  # the library path, differentiable
            return self.relu_fn(self.norm_fn(self.conv_fn(x)))
        # convolution without its bias (MIOpen / hipBLASLt), then bias + BatchNorm + activation in ONE in-place kernel
        from .restormer_ops import channel_affine_act_
        # 1x1 / small dilated or strided: deterministic GEMMs (1x1: epilogue inside the kernel on the small maps); else MIOpen + one pass
        return conv_no_bias_or(self.conv_fn, x, False, epilogue=epi)


class Conv1dNormRelu(_ConvNormRelu):
    dims = 1

    def forward(self, x):
        if x.is_cuda and x.dim() == 3 and _inference_only(x, *self.parameters()):
            y = fused_mlp1d(x, [self])  # one launch instead of GEMM + epilogue (None: a shape the kernel is not built for)
            if y is not None:
                return y
        return super().forward(x)


class Conv2dNormRelu(_ConvNormRelu):
    dims = 2

    def forward(self, x):
        y = pointwise_chain([self], x) if x.is_cuda else None
        return y if y is not None else super().forward(x)


def conv_chain(blocks, x):
    """blocks[-1](... blocks[0](x)) for Conv2dNormRelu blocks.  Where a block and its successor both run as im2col + GEMM
    (wants_im2col: the context network's dilated layers, every 3x3 layer of the coarsest level), the block's epilogue -- bias,
    BatchNorm, activation -- is not a pass of its own over the GEMM's output: the successor's unfold applies it to what it reads."""
    blocks = list(blocks)
    if not (x.is_cuda and all(isinstance(b, Conv2dNormRelu) for b in blocks) and _inference_only(x, *[p for b in blocks for p in b.parameters()])):
        for b in blocks:
            x = b(x)
        return x
    pending = None
    for i, b in enumerate(blocks):
        conv = b.conv_fn
        epi = b._epilogue()
        mine = epi is not None and not is_pointwise(conv) and wants_im2col(conv, x)
        if pending is not None and not mine:  # (cannot happen: a block is only deferred when its successor unfolds)
            raise RuntimeError("conv_chain: a deferred epilogue met a layer that does not unfold")
        if not mine:
            x = b(x)
            continue
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        defer = False
        if nxt is not None and nxt._epilogue() is not None and not is_pointwise(nxt.conv_fn):
            ho = _out_size(x.shape[2], conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0])
            wo = _out_size(x.shape[3], conv.kernel_size[1], conv.stride[1], conv.padding[1], conv.dilation[1])
            defer = wants_im2col(nxt.conv_fn, torch.empty((x.shape[0], conv.out_channels, ho, wo), device="meta"))
        x = im2col_conv(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, epilogue=None if defer else epi, pending=pending)
        pending = epi if defer else None
    return x


_CHAIN_MAX_POSITIONS = int(os.environ.get("RPE_CHAIN_MAX_POSITIONS", 70000))  # batch x positions: up to 72 x 120 x 8 (beyond: GEMM + epilogue)


def pointwise_chain(blocks, x):
    """One or two 1x1 Conv{1,2}dNormRelu blocks on x [B,C,...] in ONE launch of the fused point-wise MLP kernel (GEMM(s), bias,
    eval-mode BatchNorm and activation together; csrc/mlp_fused.hip) -- or None when that kernel does not apply (widths it is
    not instantiated for, training, large maps, where a library GEMM plus the epilogue pass is faster)."""
    if not x.is_cuda or x.dim() < 3 or not _inference_only(x, *[p for b in blocks for p in b.parameters()]):
        return None
    B, C = x.shape[0], x.shape[1]
    P = x.numel() // max(1, B * C)
    if x.dim() > 3 and (B * P > _CHAIN_MAX_POSITIONS or not x.is_contiguous()):
        return None
    y = fused_mlp1d(x.reshape(B, C, P), list(blocks))
    return None if y is None else y.reshape((B, y.shape[1]) + tuple(x.shape[2:]))


def run_chain(seq, x):
    """nn.Sequential of 1x1 Conv*NormRelu blocks (the fusers' ``mlps``): pairs of layers fused where the kernel has them."""
    blocks = list(seq)
    i = 0
    while i < len(blocks):
        y = pointwise_chain(blocks[i:i + 2], x) if i + 1 < len(blocks) else None
        if y is not None:
            x, i = y, i + 2
            continue
        x, i = blocks[i](x), i + 1
    return x


class _MLP(nn.Module):
    """utils.py:65-98: ModuleList ``convs`` of 1x1 ConvNormRelu blocks."""
    block = Conv1dNormRelu

    def __init__(self, in_channels, mlps, norm=None, activation="leaky_relu"):
        super().__init__()
        assert isinstance(in_channels, int) and isinstance(mlps, list)
        widths = [in_channels] + mlps
        self.convs = nn.ModuleList(self.block(a, b, norm=norm, activation=activation) for a, b in zip(widths[:-1], widths[1:]))

    def forward(self, x):
        for conv in self.convs:
            x = conv(x)
        return x


class MLP1d(_MLP):
    block = Conv1dNormRelu

    def forward(self, x, rows_xyz=None):
        """``rows_xyz`` [B,3,N]: return the result as PointConv rows [xyz | y | 0] (rpeflow_amd.pointconv.PackedRows)."""
        if x.is_cuda and x.dim() == 3 and len(self.convs) == 2 and _inference_only(x, *self.parameters()):
            y = fused_mlp1d(x, list(self.convs), rows_xyz=rows_xyz)  # both layers in one launch
            if y is not None:
                return y
        y = super().forward(x)
        if rows_xyz is not None:
            from .pointconv import pack_rows
            return pack_rows(rows_xyz, y)
        return y


class MLP2d(_MLP):
    block = Conv2dNormRelu


# ------------------------------------------------------------------ gathers
def batch_indexing_channel_first(batched_data: torch.Tensor, batched_indices: torch.Tensor):
    """utils.py:119-137.  data [B,C,N], indices [B,I1..Im] -> [B,C,I1..Im]."""
    assert batched_data.shape[0] == batched_indices.shape[0]
    _lib.require_gpu(batched_data, batched_indices, op="batch_indexing_channel_first")
    data = _f32(batched_data)
    B, C, N = data.shape
    idx = batched_indices.reshape(B, -1).to(torch.int64).contiguous()
    I = idx.shape[1]
    out = torch.empty((B, C, I), dtype=torch.float32, device=data.device)
    _launch(data, "batch_indexing_channel_first", _lib.lib().rpe_gather_channel_first,
            _ptr(data), *data.stride(), _ptr(idx), B, C, N, I, _ptr(out))
    return out.view([B, C] + list(batched_indices.shape[1:]))


def batch_indexing_channel_last(batched_data: torch.Tensor, batched_indices: torch.Tensor):
    """utils.py:101-116.  data [B,N,C] or [B,N], indices [B,I1..Im] -> [B,I1..Im,C]."""
    assert batched_data.shape[0] == batched_indices.shape[0]
    _lib.require_gpu(batched_data, batched_indices, op="batch_indexing_channel_last")
    squeeze = batched_data.dim() == 2
    data = _f32(batched_data.unsqueeze(-1) if squeeze else batched_data)
    B, N, C = data.shape
    idx = batched_indices.reshape(B, -1).to(torch.int64).contiguous()
    I = idx.shape[1]
    out = torch.empty((B, I, C), dtype=torch.float32, device=data.device)
    _launch(data, "batch_indexing_channel_last", _lib.lib().rpe_gather_channel_last,
            _ptr(data), *data.stride(), _ptr(idx), B, C, N, I, _ptr(out))
    shape = [B] + list(batched_indices.shape[1:])
    return out.view(shape) if squeeze else out.view(shape + [C])


# ------------------------------------------------------------------ 3-D interpolation / warp
def _interpolate(input_xyz, input_features, query_xyz, knn_indices, k, scale=1.0, residual=None):
    """``input_features``: [B,C,M], or a pair of such tensors standing for their channel-wise concatenation (never built);
    ``residual`` [B,C,Q]: added to the result inside the launch."""
    pair = isinstance(input_features, (list, tuple))
    feat_a, feat_b = (input_features if pair else (input_features, None))
    input_xyz, feat_a, query_xyz = _f32(input_xyz), _f32(feat_a), _f32(query_xyz)
    B, _, M = input_xyz.shape
    Ca, Q = feat_a.shape[1], query_xyz.shape[2]
    Cb, b_args = 0, (_NULL, 0, 0, 0)
    if feat_b is not None:
        feat_b = _f32(feat_b)
        assert feat_b.shape[0] == B and feat_b.shape[2] == M
        Cb, b_args = feat_b.shape[1], (_ptr(feat_b), *feat_b.stride())
    r_args = (_NULL, 0, 0, 0)
    if residual is not None:
        residual = _f32(residual)
        assert residual.shape == (B, Ca + Cb, Q)
        r_args = (_ptr(residual), *residual.stride())
    out = torch.empty((B, Ca + Cb, Q), dtype=torch.float32, device=input_xyz.device)
    knn_indices = knn_indices if knn_indices.stride(2) == 1 and knn_indices.stride(0) == Q * knn_indices.stride(1) \
        else knn_indices.contiguous()
    _launch(input_xyz, "knn_interpolation", _lib.lib().rpe_knn_interpolate,
            _ptr(input_xyz), *input_xyz.stride(), _ptr(feat_a), *feat_a.stride(), Ca, *b_args, Cb,
            _ptr(query_xyz), *query_xyz.stride(), _ptr(knn_indices), knn_indices.stride(1),
            B, M, Q, int(k), float(scale), *r_args, _ptr(out))
    return out


def knn_interpolation(input_xyz, input_features, query_xyz, k=3, knn_indices=None, return_indices=False):
    """utils.py:140-156.  [B,3,M], [B,C,M], [B,3,Q] -> [B,C,Q]: KNN kernel + one fused kernel
    (the reference: KNN + 2 gathers + norm + clamp + reciprocal + 2 reductions + multiply).
    ``input_features`` may be a PAIR of tensors ([B,Ca,M], [B,Cb,M]) standing for torch.cat(pair, dim=1): the decoder
    interpolates the coarser level's [flow | flow features] (RPEFlow_core.py:352) -- the result is [B,Ca+Cb,Q] either way.
    ``knn_indices`` [B,Q,>=k]: the k nearest inputs of every query when the caller has them -- the decoder interpolates
    level l + 1 -> l inside the recurrence and again for the final up-sampling of the same two clouds
    (RPEFlow_core.py:352, 426-430); ``return_indices``: also return them."""
    _lib.require_gpu(input_xyz, query_xyz, op="knn_interpolation")
    if knn_indices is None:
        knn_indices = k_nearest_neighbor(input_xyz, query_xyz, k)
    out = _interpolate(input_xyz, input_features, query_xyz, knn_indices, k)
    return (out, knn_indices) if return_indices else out


def backwarp_3d(xyz1, xyz2, flow12, k=3):
    """utils.py:159-169.  The "-flow12" features are negated and "xyz2 +" is added inside the interpolation kernel."""
    _lib.require_gpu(xyz1, xyz2, flow12, op="backwarp_3d")
    xyz1_warp = xyz1 + flow12
    knn_indices = k_nearest_neighbor(xyz1_warp, xyz2, k)
    return _interpolate(xyz1_warp, flow12, xyz2, knn_indices, k, scale=-1.0, residual=xyz2)


# ------------------------------------------------------------------ 2-D sampling
mesh_grid_cache = {}


def mesh_grid(n, h, w, device, channel_first=True):
    """utils.py:172-183 (kept for callers that want the grid itself; the kernels below
    generate pixel coordinates on the fly and never read it)."""
    key = "%d,%d,%d,%s,%s" % (n, h, w, device, channel_first)
    if key not in mesh_grid_cache:
        xs = torch.arange(0, w, dtype=torch.float32, device=device)[None, None, :].expand(n, h, w)
        ys = torch.arange(0, h, dtype=torch.float32, device=device)[None, :, None].expand(n, h, w)
        grid = torch.stack([xs, ys], 1)
        mesh_grid_cache[key] = grid if channel_first else grid.permute(0, 2, 3, 1)
    return mesh_grid_cache[key]


def _num_den(factor):
    """A scale factor as (multiplier, divisor): a number s -> (s, 1); a pair (num, den) stays -- fl(fl(x * num) / den), the two
    roundings of the reference's "x * num / den" (RPEFlow_core.py:363-370)."""
    if isinstance(factor, (tuple, list)):
        num, den = factor
        return float(num), float(den)
    return float(factor), 1.0


def _sample(sources, B, H, W, xy_ptr, xy_strides, P, add_grid, border, like):
    """rpe_bilinear_sample over ``sources`` = [(map [B,C,H,W], (scale_even, scale_odd) or None, subtract [B,C,P] or None), ...];
    a scale is a number or a (numerator, denominator) pair (``_num_den``)."""
    assert 1 <= len(sources) <= 4
    table, keep, C = (_lib.SampleSource * len(sources))(), [], 0
    for i, (m, scale, sub) in enumerate(sources):
        m = _f32(m)
        assert m.dim() == 4 and m.shape[0] == B and tuple(m.shape[2:]) == (H, W)
        if m.stride(3) != 1 or m.stride(2) != W:
            m = m.contiguous()
        (se, de), (so, do) = ((1.0, 1.0), (1.0, 1.0)) if scale is None else (_num_den(scale[0]), _num_den(scale[1]))
        sub_args = (None, 0, 0, 0)
        if sub is not None:
            sub = _f32(sub)
            assert sub.shape == (B, m.shape[1], P)
            sub_args = (sub.data_ptr(), *sub.stride())
        table[i] = _lib.SampleSource(m.data_ptr(), m.stride(0), m.stride(1), m.shape[1], se, so, de, do, *sub_args)
        keep += [m, sub]
        C += m.shape[1]
    out = torch.empty((B, C, P), dtype=torch.float32, device=like.device)
    _launch(like, "bilinear_sample", _lib.lib().rpe_bilinear_sample, ctypes.byref(table), len(sources), B, H, W,
            xy_ptr, *xy_strides, P, int(add_grid), int(border), _ptr(out))
    return out


def backwarp_2d(x, flow12, padding_mode):
    """utils.py:186-198.  x [B,C,H,W], flow12 [B,2,H,W]; padding 'border' or 'zeros'."""
    assert x.size()[-2:] == flow12.size()[-2:]
    if padding_mode not in ("border", "zeros"):
        raise NotImplementedError("backwarp_2d: padding_mode %r" % (padding_mode,))
    _lib.require_gpu(x, flow12, op="backwarp_2d")
    x, flow12 = _f32(x), _f32(flow12).contiguous()
    B, C, H, W = x.shape
    out = _sample([(x, None, None)], B, H, W, _ptr(flow12), (2 * H * W, H * W, 1), H * W, True, padding_mode == "border", x)
    return out.view(B, C, H, W)


def resize_frames(x, size, divisor=0.0, pair_split=False):
    """F.interpolate(x.float() / divisor, size, mode='bilinear', align_corners=True) for a uint8 or float [B,C,H,W] tensor
    in one launch (divisor 0: no division); pair_split: the 2*c channels are two frames, returned as [2B,c,*size] with
    frame 1 of every sample first -- the input preparation of RPEFlow.forward (RPEFlow.py:40-47, utils.py:227-241)."""
    _lib.require_gpu(x, op="resize_frames")
    assert x.dtype in (torch.uint8, torch.float32) and x.dim() == 4
    x = x.contiguous()
    B, C, H, W = x.shape
    shape = (2 * B, C // 2, *size) if pair_split else (B, C, *size)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    _launch(x, "resize_frames", _lib.lib().rpe_resize_frames, _ptr(x), int(x.dtype == torch.uint8), float(divisor), int(pair_split),
            B, C, H, W, int(size[0]), int(size[1]), _ptr(out))
    return out


def resize_flow2d(flow, target_h, target_w):
    """utils.py:217-224: bilinear (align_corners=True) resize of a [B,2,H,W] flow to the frame size, x / y components
    rescaled with it; one launch; the input itself when the size already matches."""
    origin_h, origin_w = flow.shape[2:]
    if target_h == origin_h and target_w == origin_w:
        return flow
    _lib.require_gpu(flow, op="resize_flow2d")
    assert flow.dim() == 4 and flow.shape[1] == 2
    flow = _f32(flow).contiguous()
    out = torch.empty((flow.shape[0], 2, target_h, target_w), dtype=torch.float32, device=flow.device)
    _launch(flow, "resize_flow2d", _lib.lib().rpe_resize_flow2d, _ptr(flow), flow.shape[0], origin_h, origin_w, int(target_h),
            int(target_w), target_w / origin_w, target_h / origin_h, _ptr(out))
    return out


def upsample2x_pair(a, b, scale_a=1.0):
    """(F.interpolate(a * scale_a, x2), F.interpolate(b, x2)), bilinear with align_corners=True, for two [B,C,h,w] tensors
    of one spatial size in one launch: the coarse flow (x2) and its features on their way to the next finer level
    (RPEFlow_core.py:364-369)."""
    _lib.require_gpu(a, b, op="upsample2x_pair")
    a, b = _f32(a).contiguous(), _f32(b).contiguous()
    assert a.shape[0] == b.shape[0] and a.shape[2:] == b.shape[2:]
    B, Ca, h, w = a.shape
    Cb = b.shape[1]
    out_a = torch.empty((B, Ca, 2 * h, 2 * w), dtype=torch.float32, device=a.device)
    out_b = torch.empty((B, Cb, 2 * h, 2 * w), dtype=torch.float32, device=a.device)
    _launch(a, "upsample2x_pair", _lib.lib().rpe_upsample2x_pair, _ptr(a), Ca, float(scale_a), _ptr(b), Cb, B, h, w,
            _ptr(out_a), _ptr(out_b))
    return out_a, out_b


def grid_sample_wrapper(feat_2d, xy):
    """utils.py:288-294.  feat_2d [B,C,H,W], xy [B,2,N] -> [B,C,N]; zeros outside the image."""
    return grid_sample_sources([(feat_2d, None, None)], xy)


def grid_sample_sources(sources, xy):
    """grid_sample_wrapper(torch.cat([m * scale ...], dim=1), xy) - torch.cat([subtract ...], dim=1) in ONE launch, without the
    concatenated map: ``sources`` = [(map [B,C_i,H,W], scale, subtract), ...] (at most four maps of one size); ``scale`` None or
    (even-channel factor, odd-channel factor) applied to the map's values as they are read, each factor a number or a
    (numerator, denominator) pair -- "x * num / den" with the reference's two roundings; ``subtract`` None or [B,C_i,N],
    taken off that map's samples.  What the 3-D correlation fuser does around its two calls (RPEFlow_core.py:103-111)."""
    first = sources[0][0]
    _lib.require_gpu(first, xy, op="grid_sample_wrapper")
    xy = _f32(xy)
    B, _, H, W = first.shape
    return _sample(sources, B, H, W, _ptr(xy), xy.stride(), xy.shape[2], False, False, first)


def project_points(xyz1, xyz2, camera_info, scale_x, scale_y):
    """project_pc2image (utils.py:260-285) of both frames' clouds + the sensor -> feature-map rescale (RPEFlow_core.py:316-324):
    [2B,2,N] (frame 1's samples, then frame 2's; [B,2,N] with ``xyz2`` None), one launch."""
    _lib.require_gpu(xyz1, op="project_pc2image")
    xyz1 = _f32(xyz1)
    B, _, N = xyz1.shape
    b_args, n = (_NULL, 0, 0, 0), B
    if xyz2 is not None:
        xyz2 = _f32(xyz2)
        assert xyz2.shape == xyz1.shape
        b_args, n = (_ptr(xyz2), *xyz2.stride()), 2 * B
    intr_args, cx, cy = (_NULL, 0), 0.0, 0.0
    if camera_info["projection_mode"] == "parallel":
        cx, cy = float(camera_info["cx"]), float(camera_info["cy"])
    else:
        intr = torch.stack([camera_info["f"], camera_info["cx"], camera_info["cy"]], dim=1).float().contiguous()
        assert intr.shape == (B, 3)
        intr_args = (_ptr(intr), 3)
    out = torch.empty((n, 2, N), dtype=torch.float32, device=xyz1.device)
    _launch(xyz1, "project_pc2image", _lib.lib().rpe_project_points, _ptr(xyz1), *xyz1.stride(), *b_args, B, N, *intr_args, cx, cy,
            float(scale_x), float(scale_y), _ptr(out))
    return out


@torch.no_grad()
def project_feat_with_nn_corr(xy, feat_2d, feat_3d, nn_indices=None, sampled_2d=None, subtract_last=None, append=None,
                              feat_3d_tail=None, tail_scale=(1.0, 1.0)):
    """utils.py:297-317.  xy [B,2,N], feat_2d [B,C2,H,W], feat_3d [B,C3,N], nn_indices [B,H*W]
    -> [B,C3+3,H,W].  Two launches (per-point rows, then per-pixel gather + correlation); the
    reference runs grid_sample over all N points, three gathers, a product, a mean and a concat.
    ``sampled_2d`` [B,C2,N] (optional, any strides): ``grid_sample_wrapper(feat_2d, xy)`` if the caller has it -- the 3-D
    fuser of the same (map, points) pair computes it anyway; the per-point bilinear taps are then not repeated.
    ``subtract_last`` [B,n,H,W]: subtracted from the last n projected channels; ``append`` [B,m,H,W]: concatenated behind the
    result (-> [B,C3+3+m,H,W]) -- the two steps the 2-D correlation fuser puts behind this call (RPEFlow_core.py:82-83), inside
    the second launch.  ``feat_3d_tail`` [B,Ct,N] with ``tail_scale`` = (even-channel factor, odd-channel factor; each a number or a
    (numerator, denominator) pair: fl(fl(x * num) / den), the reference's "x * (image_w - 1) / (sensor_w - 1)"): feat_3d is
    torch.cat([feat_3d, feat_3d_tail * scale], dim=1) without that tensor (the fuser's [cost volume | flow_3d xy in map units],
    :371-373)."""
    _lib.require_gpu(xy, feat_2d, feat_3d, op="project_feat_with_nn_corr")
    xy, feat_2d, feat_3d = _f32(xy), _f32(feat_2d).contiguous(), _f32(feat_3d)
    B, C2, H, W = feat_2d.shape
    if nn_indices is None:
        grid = mesh_grid(B, H, W, xy.device).reshape(B, 2, -1)
        nn_indices = k_nearest_neighbor(xy, grid, k=1)[..., 0]
    else:
        assert nn_indices.shape == (B, H * W)
    nn_indices = nn_indices.to(torch.int64).contiguous()
    C3a, N = feat_3d.shape[1], feat_3d.shape[2]
    tail_args, C3b = (_NULL, 0, 0, 0), 0
    if feat_3d_tail is not None:
        feat_3d_tail = _f32(feat_3d_tail)
        assert feat_3d_tail.shape[0] == B and feat_3d_tail.shape[2] == N
        C3b, tail_args = feat_3d_tail.shape[1], (_ptr(feat_3d_tail), *feat_3d_tail.stride())
    C3 = C3a + C3b
    n_sub = n_app = 0
    if subtract_last is not None:
        subtract_last = _f32(subtract_last).contiguous()
        n_sub = subtract_last.shape[1]
        assert subtract_last.shape == (B, n_sub, H, W) and n_sub <= C3
    if append is not None:
        append = _f32(append).contiguous()
        n_app = append.shape[1]
        assert append.shape == (B, n_app, H, W)
    out = torch.empty((B, C3 + 3 + n_app, H, W), dtype=torch.float32, device=feat_2d.device)
    rows = torch.empty((B, N, (C2 + 3) // 4 * 4 + (C3 + 3) // 4 * 4), dtype=torch.float32, device=feat_2d.device)  # kernel scratch
    sm_strides = (0, 0, 0)
    if sampled_2d is not None:
        sampled_2d = _f32(sampled_2d)
        assert sampled_2d.shape == (B, C2, N)
        sm_strides = sampled_2d.stride()
    _launch(feat_2d, "project_feat_with_nn_corr", _lib.lib().rpe_project_feat_nn_corr,
            _ptr(xy), *xy.stride(), _ptr(feat_2d), C2, H, W, _ptr(sampled_2d), *sm_strides, _ptr(feat_3d), *feat_3d.stride(), C3a,
            *tail_args, C3b, _num_den(tail_scale[0])[0], _num_den(tail_scale[1])[0], _num_den(tail_scale[0])[1], _num_den(tail_scale[1])[1],
            _ptr(nn_indices), _ptr(subtract_last), n_sub, _ptr(append), n_app, B, N, _ptr(rows), _ptr(out))
    return out


# ------------------------------------------------------------------ IDS transforms (utils.py:320-377)
def _ids_scales(persp, paral):
    """The five scalars of utils.py:333-343 / 355-361 as torch hands them to an fp32 tensor: computed in Python double,
    rounded to fp32 once."""
    sw = (paral["sensor_w"] - 1) / (persp["sensor_w"] - 1)
    sh = (paral["sensor_h"] - 1) / (persp["sensor_h"] - 1)
    return sw, sh, (paral["sensor_w"] - 1) / 2, (paral["sensor_h"] - 1) / 2, min(sw, sh)


def ids_forward(pcs, intrinsics, persp, paral):
    """perspect2parallel (utils.py:320-346) of every cloud in ``pcs`` [B,3*n,N] (RPEFlow.py:38, 68-69) in one launch:
    returns [n*B,3,N], cloud-major -- frame-1 clouds of the batch, then frame-2 clouds -- in the reference's CPU
    rounding (csrc/ids.hip)."""
    _lib.require_gpu(pcs, intrinsics, op="ids_forward")
    pcs, intrinsics = _f32(pcs), _f32(intrinsics)
    B, C, N = pcs.shape
    assert C % 3 == 0 and intrinsics.shape == (B, 3) and intrinsics.stride(1) == 1
    out = torch.empty((C // 3 * B, 3, N), dtype=torch.float32, device=pcs.device)
    _launch(pcs, "ids_forward", _lib.lib().rpe_ids_forward, _ptr(pcs), *pcs.stride(), _ptr(intrinsics), intrinsics.stride(0),
            B, C // 3, N, *_ids_scales(persp, paral), _ptr(out))
    return out


def ids_flow_inverse(xyz, flow, intrinsics, persp, paral):
    """parallel2perspect(xyz + flow) - parallel2perspect(xyz) (RPEFlow.py:91-93, utils.py:349-377), [B,3,N]."""
    _lib.require_gpu(xyz, flow, intrinsics, op="ids_flow_inverse")
    xyz, flow, intrinsics = _f32(xyz), _f32(flow), _f32(intrinsics)
    B, _, N = xyz.shape
    assert xyz.shape == flow.shape == (B, 3, N) and intrinsics.shape == (B, 3) and intrinsics.stride(1) == 1
    out = torch.empty((B, 3, N), dtype=torch.float32, device=xyz.device)
    _launch(xyz, "ids_flow_inverse", _lib.lib().rpe_ids_flow_inverse, _ptr(xyz), *xyz.stride(), _ptr(flow), *flow.stride(),
            _ptr(intrinsics), intrinsics.stride(0), B, N, *_ids_scales(persp, paral), _ptr(out))
    return out


# ------------------------------------------------------------------ fused point-wise MLP (csrc/mlp_fused.hip)
_MLP_PAIRS = [(1, 1), (1, 2), (2, 4), (4, 6), (6, 8), (8, 12), (8, 4)]
_MLP_SINGLE = [1, 2, 4, 6, 8, 12]
_ACT_CODE = {None: 0, "relu": 1, "leaky_relu": 2}
_MLP_MAX_WEIGHTS = (16384, 45000)  # one layer, two layers: beyond, the library GEMM + epilogue is as fast or faster (see _mlp_pack)


def _mlp_pack(blocks):
    """Kernel-layout copies of one or two Conv1dNormRelu blocks (include/rpeflow_hip.h), or None when their shapes
    are not among the instantiated ones; cached on the first block until a parameter or buffer changes."""
    first = blocks[0]
    tensors = [t for blk in blocks for t in (*blk.parameters(), *blk.buffers())]
    key = (len(blocks),) + tuple((t.data_ptr(), t._version) for t in tensors)
    slot = "_mlp_cache%d" % len(blocks)  # one slot per chain length: a two-layer miss and a one-layer hit do not evict each other
    cache = getattr(first, slot, None)
    if cache is not None and cache[0] == key:
        return cache[1]
    packed = None
    convs = [blk.conv_fn for blk in blocks]
    plain = all(all(k == 1 for k in c.kernel_size) and all(st == 1 for st in c.stride) and isinstance(c.padding, tuple)
                and all(pd == 0 for pd in c.padding) and c.groups == 1 for c in convs)  # (Conv1d or Conv2d: a 2-D map is its flattened positions)
    epis = [blk._epilogue() for blk in blocks]
    # every workgroup streams all the weights for its 16 points: worth it for the small layers only.  Measured, B = 4,
    # fused vs library GEMM + epilogue launches: 128 -> 128 -> 64 7 vs 14 us, 195 -> 128 -> 128 10 vs 14, 64 -> 64 4 vs 7,
    # 192 -> 128 and 273 -> 192 level, 192 -> 192 -> 128 (no instantiation: two launches) 31 vs 14.
    weights = sum(c.in_channels * c.out_channels for c in convs)
    if plain and all(e is not None for e in epis) and weights <= _MLP_MAX_WEIGHTS[len(blocks) - 1]:
        need = [(c.out_channels + 15) // 16 for c in convs]
        if len(blocks) == 2:
            fits = [p for p in _MLP_PAIRS if p[0] >= need[0] and p[1] >= need[1]]
            tiles = min(fits, key=lambda p: p[0] * p[1]) if fits else None
        else:
            fits = [t for t in _MLP_SINGLE if t >= need[0]]
            tiles = (fits[0], 0) if fits else None
        if tiles is not None:
            dev = convs[0].weight.device

            def pack_w(conv, t_out, k_groups):
                w = torch.zeros((16 * t_out, 16 * k_groups), dtype=torch.float32, device=dev)
                w[:conv.out_channels, :conv.in_channels] = conv.weight.detach().reshape(conv.out_channels, conv.in_channels).float()
                return w.reshape(t_out, 16, k_groups, 4, 4).permute(2, 0, 3, 1, 4).contiguous()  # [g][t][kk][o][s]

            def pack_ss(epi, t_out, c_out):
                scale, shift, _ = epi
                ss = torch.zeros((2, 16 * t_out), dtype=torch.float32, device=dev)
                ss[0, :c_out] = 1.0 if scale is None else scale
                if shift is not None:
                    ss[1, :c_out] = shift
                return ss

            c0 = convs[0].in_channels
            packed = dict(tiles=tiles, c0=c0, cout=convs[-1].out_channels,
                          w1=pack_w(convs[0], tiles[0], (c0 + 15) // 16), ss1=pack_ss(epis[0], tiles[0], convs[0].out_channels),
                          act1=_ACT_CODE[epis[0][2]], w2=None, ss2=None, act2=0)
            if len(blocks) == 2:
                packed.update(w2=pack_w(convs[1], tiles[1], tiles[0]), ss2=pack_ss(epis[1], tiles[1], convs[1].out_channels),
                              act2=_ACT_CODE[epis[1][2]])
    setattr(first, slot, (key, packed))
    return packed


def fused_mlp1d(x, blocks, rows_xyz=None):
    """One or two Conv1dNormRelu blocks applied to x [B,C0,N] in ONE launch; None when the kernel has no instantiation
    for these widths (the caller then runs the library path).  ``rows_xyz``: see MLP1d.forward."""
    w = _mlp_pack(blocks)
    if w is None or x.shape[1] != w["c0"]:
        return None
    _lib.require_gpu(x, op="fused_mlp1d")
    x = _f32(x)
    B, _, N = x.shape
    if rows_xyz is not None:
        rows_xyz = _f32(rows_xyz)
        stride = (w["cout"] + 3 + 15) // 16 * 16
        out = torch.empty((B, N, stride), dtype=torch.float32, device=x.device)
        xyz_args = (_ptr(rows_xyz), *rows_xyz.stride())
    else:
        stride = 0
        out = torch.empty((B, w["cout"], N), dtype=torch.float32, device=x.device)
        xyz_args = (_NULL, 0, 0, 0)
    _launch(x, "fused_mlp1d", _lib.lib().rpe_mlp1d_fused, _ptr(x), *x.stride(), B, w["c0"], N, _ptr(w["w1"]), _ptr(w["ss1"]), w["act1"],
            w["tiles"][0], _ptr(w["w2"]) if w["w2"] is not None else _NULL, _ptr(w["ss2"]) if w["ss2"] is not None else _NULL, w["act2"],
            w["tiles"][1], 0.1, w["cout"], int(rows_xyz is not None), stride, *xyz_args, _ptr(out))
    if rows_xyz is not None:
        from .pointconv import PackedRows
        return PackedRows(out, w["cout"])
    return out
