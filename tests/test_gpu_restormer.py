"""GPU: the Restormer-block kernels against plain PyTorch fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from rpeflow_amd.restormer_ops import channel_layernorm, dwconv3  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("B,C,H,W", [(2, 9, 7, 11), (1, 96, 36, 60), (3, 5, 1, 1), (2, 30, 2, 130), (2, 6, 7, 12), (1, 3, 1, 4), (2, 4, 9, 8)])
def test_dwconv_2d(B, C, H, W):
    torch.manual_seed(C)
    x, y = torch.randn(B, C, H, W), torch.randn(B, C, H, W)
    w, b = torch.randn(3 * C, 1, 3, 3), torch.randn(3 * C)
    ref = F.conv2d(torch.cat((x, y, y), 1), w, b, padding=1, groups=3 * C)
    got = dwconv3([x.to(DEV), y.to(DEV), y.to(DEV)], w.to(DEV), b.to(DEV)).cpu()
    assert (got - ref).abs().max() < 1e-5
    got = dwconv3([x.to(DEV)], w[:C].to(DEV)).cpu()
    assert (got - F.conv2d(x, w[:C], None, padding=1, groups=C)).abs().max() < 1e-5
    # gate: gelu(first half) * second half (restormer_arch.py:104-105)
    w2 = torch.randn(2 * C, 1, 3, 3)
    a, g = F.conv2d(torch.cat((x, y), 1), w2, None, padding=1, groups=2 * C).chunk(2, dim=1)
    got = dwconv3([x.to(DEV), y.to(DEV)], w2.to(DEV), gate=True).cpu()
    assert (got - F.gelu(a) * g).abs().max() < 1e-5


@pytest.mark.parametrize("B,C,N", [(2, 7, 33), (1, 64, 4096), (2, 3, 1), (2, 5, 4), (1, 3, 36)])
def test_dwconv_1d(B, C, N):
    torch.manual_seed(N)
    x, y = torch.randn(B, C, N), torch.randn(B, C, N)
    w = torch.randn(3 * C, 1, 3)
    ref = F.conv1d(torch.cat((x, y, y), 1), w, None, padding=1, groups=3 * C)
    got = dwconv3([x.to(DEV), y.to(DEV), y.to(DEV)], w.to(DEV)).cpu()
    assert (got - ref).abs().max() < 1e-5
    w2 = torch.randn(2 * C, 1, 3)
    a, g = F.conv1d(torch.cat((x, y), 1), w2, None, padding=1, groups=2 * C).chunk(2, dim=1)
    assert (dwconv3([x.to(DEV), y.to(DEV)], w2.to(DEV), gate=True).cpu() - F.gelu(a) * g).abs().max() < 1e-5


@pytest.mark.parametrize("shape", [(2, 32, 9, 15), (1, 192, 300), (3, 1, 5), (2, 96, 36, 60)])
def test_channel_layernorm(shape):
    torch.manual_seed(shape[1])
    x = torch.randn(*shape) * 3 + 1
    w, b = torch.rand(shape[1]) + 0.5, torch.randn(shape[1])
    view = [1, -1] + [1] * (len(shape) - 2)
    var = x.var(1, keepdim=True, unbiased=False)
    ref = (x - x.mean(1, keepdim=True)) / torch.sqrt(var + 1e-5) * w.view(view) + b.view(view)  # restormer_arch.py:60-63
    torch.testing.assert_close(channel_layernorm(x.to(DEV), w.to(DEV), b.to(DEV)).cpu(), ref, rtol=1e-5, atol=2e-5)
    ref = x / torch.sqrt(var + 1e-5) * w.view(view)  # BiasFree, :43-44 (C = 1: var = 0, outputs ~1e3)
    torch.testing.assert_close(channel_layernorm(x.to(DEV), w.to(DEV), None).cpu(), ref, rtol=1e-5, atol=2e-5)


def test_channel_layernorm_pair():
    from rpeflow_amd.restormer_ops import channel_layernorm_pair
    torch.manual_seed(3)
    x, y = torch.randn(2, 81, 9, 15).to(DEV), (torch.randn(2, 81, 9, 15) * 2 + 1).to(DEV)
    wx, bx, wy, by = (torch.randn(81).to(DEV) for _ in range(4))
    ox, oy = channel_layernorm_pair(x, wx, bx, y, wy, by)
    assert torch.equal(ox, channel_layernorm(x, wx, bx)) and torch.equal(oy, channel_layernorm(y, wy, by))
    ox, oy = channel_layernorm_pair(x, wx, None, y, wy, None)
    assert torch.equal(ox, channel_layernorm(x, wx, None)) and torch.equal(oy, channel_layernorm(y, wy, None))


@pytest.mark.parametrize("dims,norm,act", [(2, "batch_norm", "leaky_relu"), (1, None, "relu"), (2, None, None), (1, "batch_norm", None)])
def test_conv_norm_relu_fused_epilogue(dims, norm, act):
    """Conv{1,2}dNormRelu on the GPU (bias + eval BatchNorm + activation fused) against the same module on the CPU."""
    from rpeflow_amd.utils import Conv1dNormRelu, Conv2dNormRelu
    torch.manual_seed(dims)
    cls = Conv2dNormRelu if dims == 2 else Conv1dNormRelu
    m = cls(13, 22, kernel_size=3 if dims == 2 else 1, padding=1 if dims == 2 else 0, norm=norm, activation=act).eval()
    if norm:
        m.norm_fn.running_mean.normal_(); m.norm_fn.running_var.uniform_(0.5, 2.0)
        m.norm_fn.weight.data.uniform_(0.5, 1.5); m.norm_fn.bias.data.normal_()
    x = torch.randn(2, 13, 9, 14) if dims == 2 else torch.randn(2, 13, 37)
    with torch.no_grad():
        ref = m(x)
        got = m.to(DEV)(x.to(DEV)).cpu()
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dims,C,heads,spatial", [(2, 81, 1, (36, 60)), (2, 81, 3, (9, 15)), (2, 96, 2, (18, 30)), (1, 32, 1, (256,)),
                                                  (1, 192, 4, (301,)), (2, 32, 1, (144, 240)), (1, 64, 2, (4096,)), (1, 16, 1, (7,))])
def test_mutual_attention_fused(dims, C, heads, spatial):
    """Mutual_Attention{2D,3D} (restormer_arch.py:169-204, 251-283): gram + softmax + project_out folded into one
    matrix per sample on the GPU, against the plain PyTorch chain on the CPU."""
    from rpeflow_amd.model import _MutualAttention
    torch.manual_seed(C + heads)
    m = _MutualAttention(C, heads, False, dims).eval()
    with torch.no_grad():
        m.temperature.uniform_(0.5, 3.0)
        x, y, r = (torch.randn(2, C, *spatial) for _ in range(3))
        ref = m._forward_plain(x, y)
        mg = m.to(DEV)
        got = mg(x.to(DEV), y.to(DEV)).cpu()
        got_r = mg(x.to(DEV), y.to(DEV), residual=r.to(DEV)).cpu()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 2e-5 * max(scale, 1.0), (got - ref).abs().max().item()
    assert (got_r - (r + ref)).abs().max().item() < 2e-5 * max(scale, 1.0)


@pytest.mark.parametrize("B,H,W,scale", [(2, 9, 15, 4), (1, 36, 60, 4), (2, 5, 7, 8), (1, 1, 1, 2), (3, 4, 33, 2)])
def test_convex_upsample(B, H, W, scale):
    """RAFT convex up-sampling (models/utils.py:201-214) in one kernel against the PyTorch chain on the CPU."""
    from rpeflow_amd.model import convex_upsample
    torch.manual_seed(H + W)
    flow, mask = torch.randn(B, 2, H, W) * 3, torch.randn(B, 9 * scale * scale, H, W) * 2
    ref = convex_upsample(flow, mask, scale)                      # CPU tensors: the reference chain
    got = convex_upsample(flow.to(DEV), mask.to(DEV), scale).cpu()  # GPU tensors: the fused kernel
    assert got.shape == ref.shape
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------- section 8(f) rows against the IMPORTED REFERENCE's outputs
def _load_seeded(module, seed):
    from tests import inputs as I
    shapes = [(k, tuple(v.shape)) for k, v in module.state_dict().items()]
    module.load_state_dict({k: torch.from_numpy(v) for k, v in I.fill_params(shapes, seed).items()}, strict=True)
    return module.eval()


@pytest.mark.parametrize("name", ["cross_block2d", "cross_block2d_3heads", "cross_block3d", "cross_block2d_level1_c96",
                                  "cross_block2d_level1_c81", "cross_block3d_level1_c32"])
def test_cross_transformer_block_against_reference_golden(golden_dir, name):
    """CrossTransformerBlock2D/3D (restormer_arch.py:207-222, 287-302): the fused kernels (paired LayerNorm, depth-wise
    conv reading x|y|y, gram + softmax + project_out matrix, gated dwconv) against the reference module's own output on
    seeded parameters (tests/golden/make_golden.py fblocks)."""
    import os
    from rpeflow_amd.model import CrossTransformerBlock2D, CrossTransformerBlock3D
    from tests import cases as K
    c, x = K.FBLOCK_CASES[name], K.fblock_inputs(name)
    cls = CrossTransformerBlock2D if "2d" in name else CrossTransformerBlock3D
    m = _load_seeded(cls(dim=c["C"], num_heads=c["heads"]), c["seed"] + 1000).to(DEV)
    with torch.no_grad():
        got = m(torch.from_numpy(x["x"]).to(DEV), torch.from_numpy(x["y"]).to(DEV)).cpu().numpy()
    want = np.load(os.path.join(golden_dir, name + ".npz"))["out"]
    st = c.get("stride", 1)  # (the level-1 instantiations are stored on a stride-8 grid)
    got = got[:, :, ::st, ::st] if got.ndim == 4 else got[:, :, ::st]
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-5 * float(np.abs(want).max()))


@pytest.mark.parametrize("name", ["convex_upsample4", "convex_upsample8"])
def test_convex_upsample_against_reference_golden(golden_dir, name):
    """utils.py:201-214 against the reference function's output."""
    import os
    from rpeflow_amd.model import convex_upsample
    from tests import cases as K
    c, x = K.FBLOCK_CASES[name], K.fblock_inputs(name)
    got = convex_upsample(torch.from_numpy(x["flow"]).to(DEV), torch.from_numpy(x["mask"]).to(DEV), c["scale"]).cpu().numpy()
    np.testing.assert_allclose(got, np.load(os.path.join(golden_dir, name + ".npz"))["out"], rtol=1e-5, atol=1e-5)


def test_mutual_attention_with_bias():
    """bias=True (the constructor argument mirrors the reference's; the model passes False): with and without residual, B > 1."""
    from rpeflow_amd.model import _MutualAttention
    torch.manual_seed(5)
    m = _MutualAttention(24, 2, True, 2).eval()
    with torch.no_grad():
        m.project_out.bias.normal_(); m.qkv_dwconv.bias.normal_()
        x, y, r = (torch.randn(3, 24, 10, 12) for _ in range(3))
        ref = m._forward_plain(x, y)
        mg = m.to(DEV)
        got, got_r = mg(x.to(DEV), y.to(DEV)).cpu(), mg(x.to(DEV), y.to(DEV), residual=r.to(DEV)).cpu()
    assert (got - ref).abs().max() < 5e-5 and (got_r - (r + ref)).abs().max() < 5e-5


@pytest.mark.parametrize("norm", ["batch_norm", None])
@pytest.mark.parametrize("cin,cout,H,W", [(16, 32, 36, 60), (3, 16, 64, 96), (128, 192, 18, 30), (7, 9, 5, 7)])
def test_residual_block_tail_in_one_pass(norm, cin, cout, H, W):
    """ResidualBlock (pwc2d_core.py:6-25): act(BN(conv1(conv0 x)) + BN(down0 x)) with both branches' epilogues, the sum and the
    activation in one kernel (rpe_channel_affine_add_act) against the plain module chain on the CPU; small maps take the
    deterministic GEMM forms of the convolutions (rpeflow_amd.utils.conv_no_bias_or)."""
    from rpeflow_amd.model import ResidualBlock
    torch.manual_seed(cin + H)
    m = ResidualBlock(cin, cout, norm=norm).eval()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
    x = torch.randn(2, cin, H, W)
    with torch.no_grad():
        ref = m(x)
        got = m.to(DEV)(x.to(DEV)).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("B,cin,cout,H,W,stride", [(2, 16, 32, 36, 60, 2), (3, 3, 16, 65, 97, 2), (1, 128, 192, 18, 30, 2), (2, 7, 9, 5, 7, 2),
                                                    (1, 256, 20, 9, 15, 1), (2, 5, 3, 11, 4, 3)])
def test_residual_tail_kernel_against_float64(B, cin, cout, H, W, stride):
    """rpe_residual_tail: y = leaky(scale * y + shift + zscale * conv1x1(x, W0, stride)) in place, against float64 -- odd sizes
    (Ho = (H - 1) // stride + 1), channel counts off the kernel's group of 8, missing scale / shift / zscale."""
    from rpeflow_amd.restormer_ops import residual_tail_
    torch.manual_seed(B * 100 + cin)
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 1, 1) * 0.3
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.randn(B, cout, ho, wo)
    for with_affine in (True, False):
        scale = torch.rand(cout) + 0.5 if with_affine else None
        shift = torch.randn(cout) if with_affine else None
        zscale = torch.rand(cout) + 0.5 if with_affine else None
        short = torch.nn.functional.conv2d(x.double(), w.double(), stride=stride)
        ref = y.double() * (scale.double().view(1, -1, 1, 1) if with_affine else 1.0) + (shift.double().view(1, -1, 1, 1) if with_affine else 0.0)
        ref = ref + short * (zscale.double().view(1, -1, 1, 1) if with_affine else 1.0)
        ref = torch.where(ref >= 0, ref, ref * 0.1)
        got = residual_tail_(y.clone().to(DEV), scale.to(DEV) if with_affine else None, shift.to(DEV) if with_affine else None, x.to(DEV), w.to(DEV),
                             zscale.to(DEV) if with_affine else None, stride, "leaky_relu", 0.1).cpu()
        assert (got.double() - ref).abs().max() < 2e-6 * max(1.0, float(ref.abs().max())) * max(1.0, cin ** 0.5)


@pytest.mark.parametrize("norm", ["batch_norm", None])
@pytest.mark.parametrize("H,W", [(9, 15), (18, 30), (36, 60), (5, 7)])
def test_conv_chain_passes_epilogues_on_to_the_next_unfold(norm, H, W):
    """ContextNetwork2D / FlowEstimator2D on small maps: runs of im2col convolutions hand their bias / BatchNorm / LeakyReLU to
    the next layer's unfold (rpe_im2col_act) instead of a pass of their own -- against the plain module chain on the CPU, and
    the launch count shows the passes are gone."""
    from rpeflow_amd.model import ContextNetwork2D, FlowEstimator2D
    from rpeflow_amd import utils as U
    torch.manual_seed(H)
    nets = [ContextNetwork2D([34, 128, 128, 128, 96, 64, 32], [1, 2, 4, 8, 16, 1], norm=norm).eval(),
            FlowEstimator2D([40, 128, 128, 96, 64, 32], norm=norm, conv_last=True).eval()]
    for m in nets:
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.data.uniform_(0.5, 1.5)
                mod.bias.data.normal_(0, 0.2)
    for m in nets:
        cin = 34 if isinstance(m, ContextNetwork2D) else 40
        x = torch.randn(2, cin, H, W)
        with torch.no_grad():
            ref = m(x)
            got = m.to(DEV)(x.to(DEV))
        for a, b in zip(got, ref):
            assert a.shape == b.shape
            assert (a.cpu() - b).abs().max() < 3e-5 * max(1.0, float(b.abs().max()))
    # a layer between two unfolding layers has no pass of its own: count the epilogue launches of the dilated run
    calls = []
    import rpeflow_amd.restormer_ops as R
    real = R.channel_affine_act_
    R.channel_affine_act_ = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            nets[0](torch.randn(2, 34, 18, 30, device=DEV))
    finally:
        R.channel_affine_act_ = real
    assert len(calls) <= 3, "the dilated layers 2, 4, 8 of the context network still run an epilogue pass (%d passes)" % len(calls)


@pytest.mark.parametrize("dims,shape", [(2, (2, 96, 36, 60)), (2, (1, 81, 24, 40)), (2, (4, 32, 16, 68)), (1, (3, 64, 512)), (1, (2, 96, 1028)),
                                        (2, (1, 17, 5, 8)), (2, (2, 128, 8, 12)), (2, (2, 96, 144, 240))])
@pytest.mark.parametrize("bias", [False, True])
def test_gated_feed_forward_tail_in_one_launch(dims, shape, bias):
    """rpe_gdfn_tail (depth-wise conv + gelu gate + project_out + bias + residual in one launch) against the two launches it
    replaces -- rpe_dwconv3(gate) then rpe_pointwise_conv: the same arithmetic, so equal to fp32 re-association of the channel
    sum (bit-identical where the 1x1 kernel does not split its channel loop: the large maps) -- and against the plain module on
    the CPU; out of place, with a residual, and accumulated into the residual tensor itself."""
    from rpeflow_amd.model import _GatedFeedForward
    from rpeflow_amd.restormer_ops import dwconv3, gdfn_tail
    from rpeflow_amd.utils import conv_module
    torch.manual_seed(shape[1] + shape[-1])
    ffn = _GatedFeedForward(shape[1], 2.66, bias, dims).to(DEV).eval()
    x, r = torch.randn(shape, device=DEV), torch.randn(shape, device=DEV)
    with torch.no_grad():
        t = conv_module(ffn.project_in, x)
        g = dwconv3([t], ffn.dwconv.weight, ffn.dwconv.bias, gate=True)
        for res in (None, r):
            two = conv_module(ffn.project_out, g, residual=res)
            one = gdfn_tail(t, ffn.dwconv.weight, ffn.dwconv.bias, ffn.project_out.weight, ffn.project_out.bias, res, always=True)
            assert one is not None
            assert (one - two).abs().max() < 2e-6 * max(1.0, float(two.abs().max())), (one - two).abs().max().item()
            if shape[2:] == (144, 240):  # (enough workgroups for the 1x1 kernel's plain form: channel sums in the same order)
                assert torch.equal(one, two)
        acc = r.clone()
        out = gdfn_tail(t, ffn.dwconv.weight, ffn.dwconv.bias, ffn.project_out.weight, ffn.project_out.bias, acc, inplace=True, always=True)
        assert out.data_ptr() == acc.data_ptr() and torch.equal(out, one)
        assert torch.equal(ffn(x, residual=r), one if dims == 1 else two)  # the module: one launch for the point-cloud blocks, two for the maps
        assert gdfn_tail(torch.randn(1, 2 * 85, 9, 15, device=DEV), ffn.dwconv.weight, None, ffn.project_out.weight, always=True) is None  # W % 4 != 0
        assert (gdfn_tail(t, ffn.dwconv.weight, ffn.dwconv.bias, ffn.project_out.weight) is None) == (dims == 2)
        import copy
        cpu = copy.deepcopy(ffn).cpu()(x.cpu(), residual=r.cpu())
    assert (one.cpu() - cpu).abs().max() < 3e-5 * max(1.0, float(cpu.abs().max()))


def test_feed_forward_adds_its_residual_in_the_gemm():
    """_GatedFeedForward(x, residual=r) == r + _GatedFeedForward(x): project_out's GEMM carries the add (beta = 1)."""
    from rpeflow_amd.model import _GatedFeedForward
    torch.manual_seed(3)
    for dims, shape in ((2, (2, 32, 18, 30)), (1, (3, 64, 500))):
        ffn = _GatedFeedForward(shape[1], 2.66, False, dims).to(DEV).eval()
        x, r = torch.randn(shape, device=DEV), torch.randn(shape, device=DEV)
        with torch.no_grad():
            plain, fused = ffn(x), ffn(x, residual=r)
            cpu = ffn.cpu()(x.cpu())
        assert (fused - (r + plain)).abs().max() < 1e-5
        assert (plain.cpu() - cpu).abs().max() < 2e-5 * max(1.0, float(cpu.abs().max()))


@pytest.mark.parametrize("k,s,p,d,H,W,n", [(3, 1, 2, 2, 9, 15, 4), (3, 1, 8, 8, 18, 30, 4), (3, 2, 1, 1, 18, 30, 8), (3, 1, 4, 4, 72, 120, 4), (1, 2, 0, 1, 18, 30, 8),
                                           (3, 1, 1, 1, 9, 15, 4)])
def test_small_convolutions_are_deterministic_gemms(k, s, p, d, H, W, n):
    """The shapes MIOpen runs with atomic split-K accumulation (tools/experiments/conv_determinism.py) go through im2col / gather + one
    rocBLAS GEMM instead: equal to the library convolution to fp32 rounding, and bit-identical from call to call."""
    from rpeflow_amd.utils import conv_module, wants_im2col
    torch.manual_seed(k * 100 + d)
    conv = torch.nn.Conv2d(128, 96, k, s, p, d).to(DEV)
    x = torch.randn(n, 128, H, W, device=DEV)
    assert k == 1 or wants_im2col(conv, x)
    with torch.no_grad():
        outs = [conv_module(conv, x).clone() for _ in range(6)]
        ref = conv(x)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    assert (outs[0] - ref).abs().max() < 1e-4
    big = torch.randn(4, 128, 144, 240, device=DEV)
    assert not wants_im2col(torch.nn.Conv2d(128, 96, 3, 1, 2, 2), big)  # large maps: MIOpen's non-splitting kernels


def test_im2col_convolution_takes_strided_inputs():
    """rpe_im2col_act indexes a dense [B, C, H, W]: a channels_last or channel-sliced map reaching a small dilated / strided
    3x3 layer is made dense first (round-3 advisor finding: it was read as if dense) -- against F.conv2d on the same views."""
    from rpeflow_amd.utils import conv_module, wants_im2col
    torch.manual_seed(11)
    conv = torch.nn.Conv2d(64, 48, 3, 1, 2, 2).to(DEV)
    dense = torch.randn(4, 64, 18, 30, device=DEV)
    wide = torch.randn(4, 96, 18, 30, device=DEV)
    views = {"dense": dense, "channels_last": dense.contiguous(memory_format=torch.channels_last), "channel slice": wide[:, 16:80],
             "spatial slice": torch.randn(4, 64, 20, 34, device=DEV)[:, :, 1:19, 2:32], "transposed": dense.transpose(2, 3).contiguous().transpose(2, 3)}
    with torch.no_grad():
        for name, x in views.items():
            assert wants_im2col(conv, x), name
            assert name == "dense" or not x.is_contiguous(), name
            ref = F.conv2d(x.contiguous(), conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation)
            assert (conv_module(conv, x) - ref).abs().max() < 1e-4, name


@pytest.mark.parametrize("B,cin,cout,spatial", [(4, 192, 510, (9, 15)), (2, 7, 5, (3, 5)), (8, 67, 96, (500,)), (3, 130, 33, (36, 60)), (1, 4, 16, (64,)),
                                                (4, 255, 96, (18, 30)), (2, 96, 1020, (135,)), (4, 389, 273, (9, 15)), (1, 66, 17, (7,)), (2, 215, 81, (18, 30))])
@pytest.mark.parametrize("act", [None, "relu", "leaky_relu"])
def test_pointwise_conv_kernel_with_epilogue(B, cin, cout, spatial, act):
    """rpe_pointwise_conv (csrc/pointwise.hip): 1x1 convolution + per-channel scale / shift + activation (+ residual, also
    accumulated in place) in one launch against the plain PyTorch ops on the CPU; ragged channel counts (Cin % 4, Cout % 16),
    position counts that are not multiples of 4 / 64 (9 x 15 = 135), both Conv1d and Conv2d layouts.  The coarse-map cases (few
    workgroups, Cin >= 64) run the kernel whose four waves split the channel loop (pointwise_conv_ksplit_kernel)."""
    from rpeflow_amd.utils import pointwise_conv
    torch.manual_seed(cin * 7 + cout)
    x = torch.randn((B, cin) + spatial)
    w = torch.randn((cout, cin) + (1,) * len(spatial)) / cin ** 0.5
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.3
    res = torch.randn((B, cout) + spatial)
    conv = F.conv1d if len(spatial) == 1 else F.conv2d
    f = {None: lambda t: t, "relu": torch.relu, "leaky_relu": lambda t: F.leaky_relu(t, 0.1)}[act]
    shape = (1, -1) + (1,) * len(spatial)
    ref = f(conv(x, w) * scale.view(shape) + shift.view(shape))
    d = lambda t: t.to(DEV)
    got = pointwise_conv(d(x), d(w), epilogue=(d(scale), d(shift), act))
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert (got.cpu() - ref).abs().max() < tol
    # bias + residual, out of place and accumulated into the residual tensor itself
    bias = torch.randn(cout)
    ref2 = conv(x, w, bias) + res
    got2 = pointwise_conv(d(x), d(w), d(bias), residual=d(res))
    assert (got2.cpu() - ref2).abs().max() < tol
    r = d(res).contiguous()
    got3 = pointwise_conv(d(x), d(w), None, residual=r, inplace=True)
    assert got3.data_ptr() == r.data_ptr() and (got3.cpu() - (conv(x, w) + res)).abs().max() < tol
    again = pointwise_conv(d(x), d(w), epilogue=(d(scale), d(shift), act))
    assert torch.equal(got, again)  # fixed summation order
