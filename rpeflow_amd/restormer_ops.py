"""Host side of the Restormer-block kernels (csrc/restormer.hip): depth-wise 3x3 / 3-tap convolution with
fused channel concatenation and GDFN gate, channel LayerNorm, and the channel-attention core (csrc/attention.hip).
GPU tensors only."""
import ctypes

import torch

from . import _lib

_NULL = ctypes.c_void_p(0)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else _NULL


def dwconv3(inputs, weight, bias=None, gate=False):
    """Depth-wise conv (stride 1, padding 1) of cat(inputs, dim=1) with ``weight`` [C,1,3,3] (2-D inputs
    [B,Ci,H,W]) or [C,1,3] (1-D inputs [B,Ci,N]); gate: gelu(first half) * second half."""
    assert 1 <= len(inputs) <= 3
    _lib.require_gpu(*inputs, weight, op="dwconv3")
    xs = [t.contiguous().float() for t in inputs]
    two_d = xs[0].dim() == 4
    B = xs[0].shape[0]
    H, W = (xs[0].shape[2], xs[0].shape[3]) if two_d else (1, xs[0].shape[2])
    chans = [t.shape[1] for t in xs] + [0] * (3 - len(xs))
    C = sum(chans)
    assert weight.shape[0] == C and weight.shape[-1] == 3
    w = weight.detach().contiguous().float()
    b = bias.detach().contiguous().float() if bias is not None else None
    c_out = C // 2 if gate else C
    out = torch.empty((B, c_out, H, W) if two_d else (B, c_out, W), dtype=torch.float32, device=xs[0].device)
    ptrs = [_ptr(t) for t in xs] + [_NULL] * (3 - len(xs))
    with torch.cuda.device(xs[0].device):
        rc = _lib.lib().rpe_dwconv3(ptrs[0], chans[0], ptrs[1], chans[1], ptrs[2], chans[2], _ptr(w), _ptr(b),
                                    B, H, W, 3 if two_d else 1, int(gate), _ptr(out), _lib.stream_of(xs[0]))
    _lib.check(rc, "dwconv3")
    return out


def gdfn_tail(t, dw_weight, dw_bias, out_weight, out_bias=None, residual=None, inplace=False, always=False):
    """The gated feed-forward behind project_in in one launch: project_out(gelu(a) * b) (+ bias) (+ residual) with
    a, b = dwconv(t).chunk(2, 1); ``t`` [B, 2h, H, W] or [B, 2h, N].  Returns None where the caller should keep dwconv3(gate=True) +
    the 1x1 convolution: where the kernel does not apply (width not a multiple of 4, more than 128 output channels) and -- unless
    ``always`` -- for 2-D maps, where it is correct but slower than the two launches (3x3 taps: the gate is 3x the work of the
    3-tap form; level 1, C = 96: 296 against 237 us; the point-cloud blocks: 20-27 against 32 us).
    ``inplace``: accumulate into ``residual`` itself."""
    from .utils import _pw_packed_weight
    _lib.require_gpu(t, dw_weight, out_weight, op="gdfn_tail")
    two_d = t.dim() == 4
    B, C2 = t.shape[:2]
    H, W = (t.shape[2], t.shape[3]) if two_d else (1, t.shape[2])
    hidden, cout = C2 // 2, out_weight.shape[0]
    if W % 4 != 0 or cout > 128 or C2 % 2 or out_weight.numel() != cout * hidden or (two_d and not always):
        return None
    t = t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
    res = None
    if residual is not None:
        res = residual if (residual.dtype == torch.float32 and residual.is_contiguous()) else residual.float().contiguous()
    out = res if (inplace and res is not None and res.data_ptr() == residual.data_ptr()) else torch.empty(
        (B, cout) + tuple(t.shape[2:]), dtype=torch.float32, device=t.device)
    w = dw_weight.detach().contiguous().float()
    b = dw_bias.detach().contiguous().float() if dw_bias is not None else None
    ob = out_bias.detach().contiguous().float() if out_bias is not None else None
    with torch.cuda.device(t.device):
        rc = _lib.lib().rpe_gdfn_tail(_ptr(t), B, hidden, H, W, 3 if two_d else 1, _ptr(w), _ptr(b), _ptr(_pw_packed_weight(out_weight)), cout,
                                      _ptr(ob), _ptr(res), _ptr(out), _lib.stream_of(t))
    if rc == -2:  # RPE_EUNSUPPORTED (alignment)
        return None
    _lib.check(rc, "gdfn_tail")
    return out


def channel_layernorm(x, weight, bias=None, eps=1e-5):
    """LayerNorm over dim 1 of [B,C,...] (biased variance); bias None = BiasFree form (no mean subtraction)."""
    _lib.require_gpu(x, weight, op="channel_layernorm")
    x = x.contiguous().float()
    B, C = x.shape[:2]
    P = x.numel() // (B * C)
    out = torch.empty_like(x)
    w = weight.detach().contiguous().float()
    b = bias.detach().contiguous().float() if bias is not None else None
    with torch.cuda.device(x.device):
        rc = _lib.lib().rpe_channel_layernorm(_ptr(x), _ptr(w), _ptr(b), _ptr(out), _NULL, _NULL, _NULL, _NULL, B, C, P, float(eps),
                                              _lib.stream_of(x))
    _lib.check(rc, "channel_layernorm")
    return out


def channel_layernorm_pair(x, wx, bx, y, wy, by, eps=1e-5):
    """channel_layernorm of two same-shaped tensors with their own parameters in one launch."""
    _lib.require_gpu(x, y, wx, wy, op="channel_layernorm_pair")
    x, y = x.contiguous().float(), y.contiguous().float()
    assert x.shape == y.shape and (bx is None) == (by is None)
    B, C = x.shape[:2]
    P = x.numel() // (B * C)
    ox, oy = torch.empty_like(x), torch.empty_like(y)
    f = lambda t: t.detach().contiguous().float() if t is not None else None
    wx, bx, wy, by = f(wx), f(bx), f(wy), f(by)
    with torch.cuda.device(x.device):
        rc = _lib.lib().rpe_channel_layernorm(_ptr(x), _ptr(wx), _ptr(bx), _ptr(ox), _ptr(y), _ptr(wy), _ptr(by), _ptr(oy),
                                              B, C, P, float(eps), _lib.stream_of(x))
    _lib.check(rc, "channel_layernorm_pair")
    return ox, oy


def channel_affine_act_(y, scale, shift, act, slope=0.1):
    """In place on a contiguous [B,C,...] tensor: y = act(scale[c]*y + shift[c]); act in {None,'relu','leaky_relu'}."""
    _lib.require_gpu(y, op="channel_affine_act")
    assert y.is_contiguous() and y.dtype == torch.float32
    B, C = y.shape[:2]
    P = y.numel() // (B * C)
    code = {None: 0, "relu": 1, "leaky_relu": 2}[act]
    with torch.cuda.device(y.device):
        rc = _lib.lib().rpe_channel_affine_act(_ptr(y), _ptr(scale), _ptr(shift), B, C, P, code, float(slope), _lib.stream_of(y))
    _lib.check(rc, "channel_affine_act")
    return y


def channel_affine_add_act_(y, scale, shift, z, zscale, act, slope=0.1):
    """In place on y: y = act(scale[c]*y + shift[c] + zscale[c]*z); y, z contiguous [B,C,...] of one shape."""
    _lib.require_gpu(y, z, op="channel_affine_add_act")
    assert y.is_contiguous() and z.is_contiguous() and y.shape == z.shape and y.dtype == torch.float32 and z.dtype == torch.float32
    B, C = y.shape[:2]
    P = y.numel() // (B * C)
    code = {None: 0, "relu": 1, "leaky_relu": 2}[act]
    with torch.cuda.device(y.device):
        rc = _lib.lib().rpe_channel_affine_add_act(_ptr(y), _ptr(scale), _ptr(shift), _ptr(z), _ptr(zscale), B, C, P, code, float(slope),
                                                   _lib.stream_of(y))
    _lib.check(rc, "channel_affine_add_act")
    return y


def residual_tail_(y, scale, shift, x, shortcut_weight, shortcut_scale, stride, act, slope=0.1):
    """In place on y [B,Cout,Ho,Wo]: y = act(scale[c]*y + shift[c] + shortcut_scale[c] * conv1x1(x, shortcut_weight, stride)) -- the
    residual block's tail with the strided 1x1 shortcut (no padding) computed inside the pass (rpe_residual_tail)."""
    _lib.require_gpu(y, x, shortcut_weight, op="residual_tail")
    assert y.is_contiguous() and x.is_contiguous() and y.dtype == torch.float32 and x.dtype == torch.float32 and x.dim() == 4
    B, cin, H, W = x.shape
    cout = y.shape[1]
    w = shortcut_weight.reshape(cout, cin)
    assert w.is_contiguous() and w.dtype == torch.float32
    assert tuple(y.shape) == (B, cout, (H - 1) // stride + 1, (W - 1) // stride + 1), (tuple(y.shape), tuple(x.shape), stride)
    code = {None: 0, "relu": 1, "leaky_relu": 2}[act]
    with torch.cuda.device(y.device):
        rc = _lib.lib().rpe_residual_tail(_ptr(y), _ptr(scale), _ptr(shift), _ptr(x), _ptr(w), _ptr(shortcut_scale), B, cin, cout, H, W, int(stride),
                                          code, float(slope), _lib.stream_of(y))
    _lib.check(rc, "residual_tail")
    return y


def channel_attention_matrix(qkv, heads, temperature, w_out, eps=1e-12, packed=False):
    """qkv [B,3C,...] contiguous (q | k | v along channels).  Returns M [B,C,C] with
    project_out(softmax(normalize(q) normalize(k)^T * temperature) v) == M @ v  (restormer_arch.py:184-203).
    ``packed``: M[b] in rpe_pointwise_conv's weight-fragment order [B, ceil(C/16), ceil(C/4), 64] (for attention_apply)."""
    _lib.require_gpu(qkv, temperature, w_out, op="channel_attention_matrix")
    assert qkv.is_contiguous() and qkv.dtype == torch.float32 and qkv.shape[1] % (3 * heads) == 0
    B, C = qkv.shape[0], qkv.shape[1] // 3
    P = qkv.numel() // (B * 3 * C)
    c = C // heads
    t = temperature.detach().reshape(-1).contiguous().float()
    w = w_out.detach().reshape(C, C).contiguous().float()
    assert t.numel() == heads
    L = _lib.lib()
    ws = torch.empty(L.rpe_channel_attention_workspace_floats(B, heads, c, P), dtype=torch.float32, device=qkv.device)
    m = torch.empty((B, (C + 15) // 16, (C + 3) // 4, 64) if packed else (B, C, C), dtype=torch.float32, device=qkv.device)
    q_ptr = qkv.data_ptr()
    with torch.cuda.device(qkv.device):
        rc = L.rpe_channel_attention_matrix(ctypes.c_void_p(q_ptr), ctypes.c_void_p(q_ptr + 4 * C * P), 3 * C * P, _ptr(t), _ptr(w),
                                            B, heads, c, P, float(eps), _ptr(ws), _ptr(m), int(bool(packed)), _lib.stream_of(qkv))
    _lib.check(rc, "channel_attention_matrix")
    return m


def attention_apply(qkv, m_packed, residual=None, bias=None):
    """out[b] = m[b] @ v[b] (+ bias) (+ residual): v = the last third of qkv's channels (read in place, batch stride 3 C P), m
    from channel_attention_matrix(..., packed=True); one launch of the 1x1 kernel with per-sample weights."""
    _lib.require_gpu(qkv, m_packed, op="attention_apply")
    B, C = qkv.shape[0], qkv.shape[1] // 3
    P = qkv.numel() // (B * 3 * C)
    res = None if residual is None else (residual if (residual.dtype == torch.float32 and residual.is_contiguous()) else residual.float().contiguous())
    out = torch.empty((B, C) + tuple(qkv.shape[2:]), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        rc = _lib.lib().rpe_pointwise_conv(ctypes.c_void_p(qkv.data_ptr() + 4 * 2 * C * P), 3 * C * P, B, C, P, _ptr(m_packed),
                                           m_packed[0].numel(), C, _NULL, _ptr(bias.float().contiguous()) if bias is not None else _NULL,
                                           0, 0.1, _ptr(res) if res is not None else _NULL, C * P, _ptr(out), _lib.stream_of(qkv))
    _lib.check(rc, "attention_apply")
    return out


def convex_upsample(flow, mask, scale_factor):
    """flow [B,2,H,W], mask [B,9*s*s,H,W] -> [B,2,H*s,W*s] (models/utils.py:201-214) in one kernel."""
    _lib.require_gpu(flow, mask, op="convex_upsample")
    flow, mask = flow.contiguous().float(), mask.contiguous().float()
    B, _, H, W = flow.shape
    assert flow.shape[1] == 2 and mask.shape == (B, 9 * scale_factor * scale_factor, H, W)
    out = torch.empty((B, 2, H * scale_factor, W * scale_factor), dtype=torch.float32, device=flow.device)
    with torch.cuda.device(flow.device):
        rc = _lib.lib().rpe_convex_upsample(_ptr(flow), _ptr(mask), B, H, W, int(scale_factor), _ptr(out), _lib.stream_of(flow))
    _lib.check(rc, "convex_upsample")
    return out
