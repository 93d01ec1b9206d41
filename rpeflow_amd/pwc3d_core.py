"""3-D pyramid blocks -- mirror of the reference's models/pwc3d_core.py.

Same classes, constructor arguments, forward signatures and state-dict keys.
KNN / FPS / gathers / PointConv grouping run as HIP kernels; the dense 1x1
convolutions stay on PyTorch (MIOpen / hipBLASLt).
"""
import torch
import torch.nn as nn

from .csrc import furthest_point_sampling, k_nearest_neighbor
from .csrc.wrapper import k_nearest_neighbor_multi
from .pointconv import PointConvDownSampling, PointConvNoSampling
from .utils import MLP1d, MLP2d, batch_indexing_channel_first


_LEVEL0_INDEX = {}  # constants of the pyramid by (batch, points, device): the level-0 index; the zero point of the level-0 feature


def build_pc_pyramid(pc1, pc2, n_samples_list, sample_index_both=None, return_both=False):
    """pwc3d_core.py:8-28: one FPS over both clouds, every level a prefix of its order.
    ``sample_index_both`` ([2B, >=max(n_samples_list)] int64): that order, when the caller computed it already.
    ``pc1`` / ``pc2`` that are the two halves of ONE [2B,3,N] tensor (what the IDS transform returns) are sampled where they
    lie -- no torch.cat -- and the levels of both clouds come out of ONE gather; ``return_both``: also the list of the
    stacked levels ([2B,3,n], frame 1 first: views of that gather's output)."""
    batch_size, _, n_points = pc1.shape
    pc_both = _stacked(pc1, pc2)
    if pc_both is None:
        pc_both = torch.cat([pc1, pc2], dim=0)
    if sample_index_both is None:
        sample_index_both = furthest_point_sampling(pc_both.transpose(1, 2), max(n_samples_list))
    sample_index1, sample_index2 = sample_index_both[:batch_size], sample_index_both[batch_size:]

    key = (batch_size, n_points, pc1.device, torch.is_inference_mode_enabled())
    lv0_index = _LEVEL0_INDEX.get(key)
    if lv0_index is None:  # a constant: made once (kept only when made outside a stream capture, whose memory belongs to the graph)
        lv0_index = torch.arange(n_points, device=pc1.device)[None, :].expand(batch_size, n_points)
        if not (pc1.is_cuda and torch.cuda.is_current_stream_capturing()):
            _LEVEL0_INDEX[key] = lv0_index
    xyzs1, xyzs2, sample_indices1, sample_indices2, both = [pc1], [pc2], [lv0_index], [lv0_index], [pc_both]
    # one gather at the largest size for both clouds; smaller levels are prefixes of it (pwc3d_core.py:22-26)
    n_max = max(n_samples_list)
    top = batch_indexing_channel_first(pc_both, sample_index_both[:, :n_max])
    for n_samples in n_samples_list:
        sample_indices1.append(sample_index1[:, :n_samples])
        sample_indices2.append(sample_index2[:, :n_samples])
        xyzs1.append(top[:batch_size, :, :n_samples])
        xyzs2.append(top[batch_size:, :, :n_samples])
        both.append(top[:, :, :n_samples])
    if return_both:
        return xyzs1, xyzs2, sample_indices1, sample_indices2, both
    return xyzs1, xyzs2, sample_indices1, sample_indices2


def constants_intact():
    """The cached pyramid constants still hold their values: the level-0 index is 0 .. N-1 in every row (an expanded view --
    PyTorch refuses in-place writes through it -- handed out in ``sample_indices`` as the reference hands out its own,
    pwc3d_core.py:17-19), the zero point is zero."""
    for key, t in _LEVEL0_INDEX.items():
        if key[0] == "zero point":
            if int(t.count_nonzero()) != 0:
                return False
        elif not torch.equal(t[0], torch.arange(t.shape[1], device=t.device)):
            return False
    return True


def clear_constants():
    _LEVEL0_INDEX.clear()


def _stacked(a, b):
    """The tensor [a; b] (stacked on dim 0) if ``a`` and ``b`` are its two halves already, else None."""
    if (a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype and a.device == b.device and a._base is not None
            and a._base is b._base and a._base.dim() == a.dim() and a._base.shape[0] == 2 * a.shape[0] and a._base.shape[1:] == a.shape[1:]
            and a._base.stride() == a.stride() and a.data_ptr() == a._base.data_ptr()
            and b.data_ptr() == a.data_ptr() + a.shape[0] * a.stride(0) * a.element_size()):
        return a._base
    return None


class FeaturePyramid3D(nn.Module):
    """pwc3d_core.py:31-57."""

    def __init__(self, n_channels, norm=None, k=16):
        super().__init__()
        self.level0_mlp = MLP1d(3, [n_channels[0], n_channels[0]])
        self.pyramid_mlps = nn.ModuleList()
        self.pyramid_convs = nn.ModuleList()
        for i in range(len(n_channels) - 1):
            self.pyramid_mlps.append(MLP1d(n_channels[i], [n_channels[i], n_channels[i + 1]]))
            self.pyramid_convs.append(PointConvDownSampling(n_channels[i + 1], n_channels[i + 1], norm=norm, k=k))

    def forward(self, xyzs):
        assert len(xyzs) == len(self.pyramid_mlps) + 1
        # the layers' neighbour searches (pointconv.py:46) need the coordinates only: all levels in one launch up front
        knns = k_nearest_neighbor_multi([(xyzs[i], xyzs[i + 1]) for i in range(len(xyzs) - 1)], self.pyramid_convs[0].k)
        # level 0: the MLP of an all-zero input (pwc3d_core.py:51-52), i.e. ONE vector for every point (eval-mode BatchNorm
        # is per point): computed on a single point and broadcast -- a stride-0 view the next layer's kernel reads as it is
        if xyzs[0].is_cuda and not self.training and not torch.is_grad_enabled():
            zero_key = ("zero point", xyzs[0].device, xyzs[0].dtype, torch.is_inference_mode_enabled())
            zero = _LEVEL0_INDEX.get(zero_key)
            if zero is None:  # (a constant, like the level-0 index: kept unless made inside a stream capture)
                zero = xyzs[0].new_zeros((1, 3, 1))
                if not torch.cuda.is_current_stream_capturing():
                    _LEVEL0_INDEX[zero_key] = zero
            feats = [self.level0_mlp(zero).expand(xyzs[0].shape[0], -1, xyzs[0].shape[2])]
        else:
            feats = [self.level0_mlp(torch.zeros_like(xyzs[0]))]
        for i in range(len(xyzs) - 1):
            # the MLP writes the PointConv layer's gather rows [xyz | features] directly: no packing pass
            feats.append(self.pyramid_convs[i](xyzs[i], self.pyramid_mlps[i](feats[-1], rows_xyz=xyzs[i]), xyzs[i + 1], knn_indices=knns[i]))
        return feats


def _pad_channels(c):
    """Channel counts the Correlation3D kernels are instantiated for (16 * T, T in 1, 2, 4, 6, 8, 12); None: wider than that."""
    for cp in (16, 32, 64, 96, 128, 192):
        if c <= cp:
            return cp
    return None


class Correlation3D(nn.Module):
    """pwc3d_core.py:60-117.

    The first cost_mlp layer is linear in the concatenation [feat1 | feat2_nbr | rel]
    (pwc3d_core.py:92-94), so its feat1 / feat2 blocks are applied per POINT before the
    gather -- ``project()``: one batched GEMM, which depends on the features only and can
    be issued long before the warped cloud exists -- instead of per (point, neighbour)
    pair, and the [B, 2C+3, N, k] tensor is never built.  Behind the neighbour search the
    rest is two kernels (csrc/corr3d_fused.hip): gather + rel block + second layer on MFMA +
    weight_net2 + k-sum, then the second hop.  Results differ from the reference by fp32
    re-association only.  The kernels are built for k = 16 and up to 192 output channels (every shipped
    configuration); any other k or width, and inputs that require grad, take the reference's own op
    sequence on the GPU (HIP neighbour search and gathers, library convolutions)."""

    def __init__(self, in_channels, out_channels, k=16):
        super().__init__()
        self.k = k
        self.fusable = k == 16 and _pad_channels(out_channels) is not None
        self.cost_mlp = MLP2d(3 + 2 * in_channels, [out_channels, out_channels], activation="leaky_relu")
        self.weight_net1 = MLP2d(3, [8, 8, out_channels], activation="relu")
        self.weight_net2 = MLP2d(3, [8, 8, out_channels], activation="relu")
        self._cache = None

    def _weights(self):
        """fp32 copies of the parameters in the layouts the kernels read (include/rpeflow_hip.h); rebuilt when a
        parameter changes (load_state_dict, .to(), an optimiser step)."""
        params = list(self.parameters())
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._cache is None or self._cache[0] != key:
            first, second = self.cost_mlp.convs[0].conv_fn, self.cost_mlp.convs[1].conv_fn
            w = first.weight.detach()[:, :, 0, 0].float()
            c_out, c_in = w.shape[0], (w.shape[1] - 3) // 2
            cp, dev = _pad_channels(c_out), w.device
            t = cp // 16

            def pad(x, *shape):
                out = torch.zeros(shape, dtype=torch.float32, device=dev)
                out[tuple(slice(0, n) for n in x.shape)] = x
                return out

            def net(m):
                c0, c1, c2 = (conv.conv_fn for conv in m.convs)
                w3 = pad(c2.weight.detach().float().reshape(c_out, 8), cp, 8)
                return [c0.weight.detach().float().reshape(8, 3).contiguous(), c0.bias.detach().float().contiguous(),
                        c1.weight.detach().float().reshape(8, 8).contiguous(), c1.bias.detach().float().contiguous(),
                        w3.reshape(t, 16, 2, 4).permute(0, 3, 1, 2).contiguous(),  # [t][n][s][kk] -> [t][kk][n][s]
                        pad(c2.bias.detach().float(), cp)]

            w2 = pad(second.weight.detach()[:, :, 0, 0].float(), cp, cp)
            self._cache = (key, dict(
                c_out=c_out, c_in=c_in, cp=cp,
                # per-point halves of the first layer for the stacked [feat1; feat2] batch: columns padded to cp
                w_ab_t=torch.stack([pad(w[:, :c_in].t(), c_in, cp), pad(w[:, c_in:2 * c_in].t(), c_in, cp)]),
                bias_ab=torch.stack([pad(first.bias.detach().float(), cp), torch.zeros(cp, device=dev)]),
                wc4=pad(w[:, 2 * c_in:], cp, 4),
                w2p=w2.reshape(t, 16, t, 4, 4).permute(2, 0, 3, 1, 4).contiguous(),  # [t][n][g][kk][s] -> [g][t][kk][n][s]
                b2=pad(second.bias.detach().float(), cp),
                net1=net(self.weight_net1), net2=net(self.weight_net2)))
        return self._cache[1]

    def project_stacked(self, feat_both):
        """project() for the two clouds' features stacked on the batch axis ([2B,C,N], cloud 1 first): ONE batched GEMM.
        None where the fused kernels do not apply (forward() then runs the reference's op sequence and needs no projection)."""
        if not self.fusable:
            return None
        w = self._weights()
        batch_size = feat_both.shape[0] // 2
        if w.get("stacked_for") == batch_size:  # per-sample copies of the two weight blocks: constants of the batch size
            wt, bias = w["stacked_wt"], w["stacked_bias"]
        else:
            wt = w["w_ab_t"].repeat_interleave(batch_size, dim=0)
            bias = w["bias_ab"].repeat_interleave(batch_size, dim=0)[:, None, :]
            if not (wt.is_cuda and torch.cuda.is_current_stream_capturing()):  # (a capturing graph owns what is allocated inside it)
                w["stacked_wt"], w["stacked_bias"], w["stacked_for"] = wt, bias, batch_size
        rows = torch.baddbmm(bias, feat_both.float().transpose(1, 2), wt)
        return rows[:batch_size], rows[batch_size:]

    def project(self, feat1, feat2):
        """(p1_rows [B,N,Cp], p2_rows [B,M,Cp]): the feat1 / feat2 blocks of cost_mlp's first layer per point, channel-last.
        Depends on the features only; callers that know them early pass the result to forward(projected=...)."""
        if not self.fusable:
            return None
        if feat1.shape[2] == feat2.shape[2]:
            both = _stacked(feat1, feat2)  # (the two frames' features usually ARE the halves of one [2B,C,N] tensor)
            return self.project_stacked(both if both is not None else torch.cat([feat1, feat2], dim=0))
        w = self._weights()
        batch_size = feat1.shape[0]
        p1 = torch.baddbmm(w["bias_ab"][0][None, None, :], feat1.float().transpose(1, 2), w["w_ab_t"][0][None].expand(batch_size, -1, -1))
        p2 = torch.matmul(feat2.float().transpose(1, 2), w["w_ab_t"][1])
        return p1, p2

    def _general(self, xyz1, feat1, xyz2, feat2, knn_indices_1in1):
        """pwc3d_core.py:81-117 op for op."""
        from .pointconv import gather_channel_first as gather
        batch_size, in_channels, n_points = feat1.shape
        knn_1in2 = k_nearest_neighbor(input_xyz=xyz2, query_xyz=xyz1, k=self.k)
        rel2 = gather(xyz2, knn_1in2) - xyz1.view(batch_size, 3, n_points, 1)
        cat = torch.cat([feat1[:, :, :, None].expand(batch_size, in_channels, n_points, self.k), gather(feat2, knn_1in2), rel2], dim=1)
        p2n_cost = torch.sum(self.weight_net2(rel2) * self.cost_mlp(cat), dim=3)
        if knn_indices_1in1 is not None:
            assert knn_indices_1in1.shape == torch.Size([batch_size, n_points, self.k])
        else:
            knn_indices_1in1 = k_nearest_neighbor(input_xyz=xyz1, query_xyz=xyz1, k=self.k)
        rel1 = gather(xyz1, knn_indices_1in1) - xyz1.view(batch_size, 3, n_points, 1)
        return torch.sum(self.weight_net1(rel1) * gather(p2n_cost, knn_indices_1in1), dim=3)

    def forward(self, xyz1, feat1, xyz2, feat2, knn_indices_1in1=None, projected=None):
        from . import _lib
        from .utils import _inference_only, _ptr
        _lib.require_gpu(xyz1, feat1, xyz2, feat2, op="Correlation3D")
        if not self.fusable or not _inference_only(xyz1, feat1, xyz2, feat2, *self.parameters()):
            return self._general(xyz1, feat1, xyz2, feat2, knn_indices_1in1)
        batch_size, _, n_points = feat1.shape
        m_points = xyz2.shape[2]
        w = self._weights()
        xyz1, xyz2 = xyz1.float(), xyz2.float()
        lib, stream = _lib.lib(), _lib.stream_of(xyz1)

        knn_indices_1in2 = k_nearest_neighbor(input_xyz=xyz2, query_xyz=xyz1, k=self.k)
        p1_rows, p2_rows = projected if projected is not None else self.project(feat1, feat2)
        p2n_rows = torch.empty((batch_size, n_points, w["cp"]), dtype=torch.float32, device=xyz1.device)
        with torch.cuda.device(xyz1.device):
            rc = lib.rpe_corr3d_cost(_ptr(p1_rows), _ptr(p2_rows), _ptr(w["wc4"]), _ptr(w["w2p"]), _ptr(w["b2"]),
                                     *[_ptr(t) for t in w["net2"]], _ptr(xyz1), *xyz1.stride(), _ptr(xyz2), *xyz2.stride(),
                                     _ptr(knn_indices_1in2), knn_indices_1in2.stride(1), batch_size, w["cp"], n_points, m_points,
                                     0.1, _ptr(p2n_rows), stream)
        _lib.check(rc, "Correlation3D (point-to-neighbour cost)")

        if knn_indices_1in1 is not None:
            assert knn_indices_1in1.shape == torch.Size([batch_size, n_points, self.k])
            knn_indices_1in1 = knn_indices_1in1.to(torch.int64).contiguous()
        else:
            knn_indices_1in1 = k_nearest_neighbor(input_xyz=xyz1, query_xyz=xyz1, k=self.k)
        n2n_cost = torch.empty((batch_size, w["c_out"], n_points), dtype=torch.float32, device=xyz1.device)
        with torch.cuda.device(xyz1.device):
            rc = lib.rpe_corr3d_n2n(_ptr(p2n_rows), *[_ptr(t) for t in w["net1"]], _ptr(xyz1), *xyz1.stride(),
                                    _ptr(knn_indices_1in1), knn_indices_1in1.stride(1), batch_size, w["c_out"], w["cp"], n_points,
                                    _ptr(n2n_cost), stream)
        _lib.check(rc, "Correlation3D (neighbour-to-neighbour cost)")
        return n2n_cost


class FlowEstimator3D(nn.Module):
    """pwc3d_core.py:120-148."""
    concatenates = True  # forward() accepts the list of tensors a caller would torch.cat

    def __init__(self, n_channels, norm=None, conv_last=True, k=16):
        super().__init__()
        self.point_conv1 = PointConvNoSampling(in_channels=n_channels[0], out_channels=n_channels[1], norm=norm, k=k)
        self.point_conv2 = PointConvNoSampling(in_channels=n_channels[1], out_channels=n_channels[2], norm=norm, k=k)
        self.mlp = MLP1d(n_channels[2], [n_channels[2], n_channels[3]])
        self.conv_last = nn.Conv1d(n_channels[3], 3, kernel_size=1) if conv_last else None

    def forward(self, xyz, feat, knn_indices):
        """``feat``: [B,C,N], or the list of tensors the caller would concatenate along the channels
        (RPEFlow_core.py:382-391): the first layer's packing pass does that concatenation."""
        rows = self.point_conv1.forward(xyz, feat, knn_indices, out_rows=True)  # channel-last, the second layer's gather source
        feat = self.point_conv2.forward(xyz, rows, knn_indices)
        feat = self.mlp(feat)
        if self.conv_last is not None:
            from .utils import conv_module
            return feat, conv_module(self.conv_last, feat)
        return feat
