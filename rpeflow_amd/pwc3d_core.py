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


def build_pc_pyramid(pc1, pc2, n_samples_list, sample_index_both=None):
    """pwc3d_core.py:8-28: one FPS over both clouds, every level a prefix of its order.
    ``sample_index_both`` ([2B, >=max(n_samples_list)] int64): that order, when the caller computed it already."""
    batch_size, _, n_points = pc1.shape
    if sample_index_both is None:
        pc_both = torch.cat([pc1, pc2], dim=0)
        sample_index_both = furthest_point_sampling(pc_both.transpose(1, 2), max(n_samples_list))
    sample_index1, sample_index2 = sample_index_both[:batch_size], sample_index_both[batch_size:]

    lv0_index = torch.arange(n_points, device=pc1.device)[None, :].expand(batch_size, n_points)
    xyzs1, xyzs2, sample_indices1, sample_indices2 = [pc1], [pc2], [lv0_index], [lv0_index]
    # one gather at the largest size; smaller levels are prefixes of it (pwc3d_core.py:22-26)
    n_max = max(n_samples_list)
    top1 = batch_indexing_channel_first(pc1, sample_index1[:, :n_max])
    top2 = batch_indexing_channel_first(pc2, sample_index2[:, :n_max])
    for n_samples in n_samples_list:
        sample_indices1.append(sample_index1[:, :n_samples])
        sample_indices2.append(sample_index2[:, :n_samples])
        xyzs1.append(top1[:, :, :n_samples])
        xyzs2.append(top2[:, :, :n_samples])
    return xyzs1, xyzs2, sample_indices1, sample_indices2


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
        feats = [self.level0_mlp(torch.zeros_like(xyzs[0]))]
        for i in range(len(xyzs) - 1):
            feats.append(self.pyramid_convs[i](xyzs[i], self.pyramid_mlps[i](feats[-1]), xyzs[i + 1], knn_indices=knns[i]))
        return feats


class Correlation3D(nn.Module):
    """pwc3d_core.py:60-117.

    The first cost_mlp layer is linear in the concatenation [feat1 | feat2_nbr | rel]
    (pwc3d_core.py:92-94), so it is applied per POINT before the gather --
    W_a feat1 + gather(W_b feat2) + W_c rel -- instead of per (point, neighbour) pair:
    k = 16 times fewer multiply-adds in that layer, and the [B, 2C+3, N, k] tensor is
    never built.  The gather + sum + leaky_relu, both weight nets and both k-sums run
    in three launches of two HIP kernels (csrc/corr3d.hip); the only dense work left is
    three GEMMs on hipBLASLt.  Results differ from the reference by fp32 re-association
    only."""

    def __init__(self, in_channels, out_channels, k=16):
        super().__init__()
        if k != 16:
            raise NotImplementedError("rpeflow_amd Correlation3D is built for k=16 (conf/*/*.yaml pwc3d.k)")
        self.k = k
        self.cost_mlp = MLP2d(3 + 2 * in_channels, [out_channels, out_channels], activation="leaky_relu")
        self.weight_net1 = MLP2d(3, [8, 8, out_channels], activation="relu")
        self.weight_net2 = MLP2d(3, [8, 8, out_channels], activation="relu")
        self._cache = None

    def _weights(self, in_channels):
        """Contiguous fp32 views of the parameters in the layout the kernels read; rebuilt when a
        parameter changes (load_state_dict, .to(), an optimiser step)."""
        params = list(self.parameters())
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._cache is None or self._cache[0] != key:
            first, second = self.cost_mlp.convs[0].conv_fn, self.cost_mlp.convs[1].conv_fn
            w = first.weight.detach()[:, :, 0, 0].float()
            c = in_channels
            net = lambda m: [t.detach().float().reshape(t.shape[0], -1).contiguous() if t.dim() > 1 else t.detach().float().contiguous()
                             for conv in m.convs for t in (conv.conv_fn.weight, conv.conv_fn.bias)]
            self._cache = (key, dict(
                w_a=w[:, :c].contiguous(), w_b=w[:, c:2 * c].contiguous(), w_c=w[:, 2 * c:].contiguous(),
                b_first=first.bias.detach().float().contiguous(),
                w_second=second.weight.detach()[:, :, 0, 0].float().contiguous(),
                b_second=second.bias.detach().float().contiguous(),
                net1=net(self.weight_net1), net2=net(self.weight_net2)))
        return self._cache[1]

    def forward(self, xyz1, feat1, xyz2, feat2, knn_indices_1in1=None):
        from . import _lib
        from .utils import _ptr
        _lib.require_gpu(xyz1, feat1, xyz2, feat2, op="Correlation3D")
        batch_size, in_channels, n_points = feat1.shape
        m_points = xyz2.shape[2]
        w = self._weights(in_channels)
        c_out = w["w_a"].shape[0]
        xyz1, xyz2 = xyz1.float(), xyz2.float()
        lib, stream = _lib.lib(), _lib.stream_of(xyz1)

        knn_indices_1in2 = k_nearest_neighbor(input_xyz=xyz2, query_xyz=xyz1, k=self.k)
        part1 = torch.baddbmm(w["b_first"][None, :, None], w["w_a"][None].expand(batch_size, -1, -1), feat1.float())
        part2 = torch.matmul(w["w_b"], feat2.float())
        hidden = torch.empty((batch_size, c_out, n_points, self.k), dtype=torch.float32, device=xyz1.device)
        with torch.cuda.device(xyz1.device):
            rc = lib.rpe_corr3d_hidden(_ptr(part1), _ptr(part2), _ptr(w["w_c"]), _ptr(xyz1), *xyz1.stride(),
                                       _ptr(xyz2), *xyz2.stride(), _ptr(knn_indices_1in2), knn_indices_1in2.stride(1),
                                       batch_size, c_out, n_points, m_points, 0.1, _ptr(hidden), stream)
        _lib.check(rc, "Correlation3D (hidden)")
        # second cost_mlp layer: one GEMM over all (point, neighbour) pairs, then leaky_relu(0.1)
        from .restormer_ops import channel_affine_act_
        p2p_cost = torch.matmul(w["w_second"], hidden.view(batch_size, c_out, -1))
        p2p_cost = channel_affine_act_(p2p_cost, None, w["b_second"], "leaky_relu", 0.1)  # bias + leaky_relu in one in-place pass

        p2n_cost = torch.empty((batch_size, c_out, n_points), dtype=torch.float32, device=xyz1.device)
        with torch.cuda.device(xyz1.device):
            rc = lib.rpe_corr3d_weighted_sum(_ptr(p2p_cost), 0, *[_ptr(t) for t in w["net2"]], _ptr(xyz1), *xyz1.stride(),
                                             _ptr(xyz2), *xyz2.stride(), _ptr(knn_indices_1in2), knn_indices_1in2.stride(1),
                                             batch_size, c_out, n_points, m_points, _ptr(p2n_cost), stream)
        _lib.check(rc, "Correlation3D (p2n)")

        if knn_indices_1in1 is not None:
            assert knn_indices_1in1.shape == torch.Size([batch_size, n_points, self.k])
            knn_indices_1in1 = knn_indices_1in1.to(torch.int64).contiguous()
        else:
            knn_indices_1in1 = k_nearest_neighbor(input_xyz=xyz1, query_xyz=xyz1, k=self.k)
        n2n_cost = torch.empty_like(p2n_cost)
        with torch.cuda.device(xyz1.device):
            rc = lib.rpe_corr3d_weighted_sum(_ptr(p2n_cost), 1, *[_ptr(t) for t in w["net1"]], _ptr(xyz1), *xyz1.stride(),
                                             _ptr(xyz1), *xyz1.stride(), _ptr(knn_indices_1in1), knn_indices_1in1.stride(1),
                                             batch_size, c_out, n_points, n_points, _ptr(n2n_cost), stream)
        _lib.check(rc, "Correlation3D (n2n)")
        return n2n_cost


class FlowEstimator3D(nn.Module):
    """pwc3d_core.py:120-148."""

    def __init__(self, n_channels, norm=None, conv_last=True, k=16):
        super().__init__()
        self.point_conv1 = PointConvNoSampling(in_channels=n_channels[0], out_channels=n_channels[1], norm=norm, k=k)
        self.point_conv2 = PointConvNoSampling(in_channels=n_channels[1], out_channels=n_channels[2], norm=norm, k=k)
        self.mlp = MLP1d(n_channels[2], [n_channels[2], n_channels[3]])
        self.conv_last = nn.Conv1d(n_channels[3], 3, kernel_size=1) if conv_last else None

    def forward(self, xyz, feat, knn_indices):
        feat = self.point_conv1.forward(xyz, feat, knn_indices)
        feat = self.point_conv2.forward(xyz, feat, knn_indices)
        feat = self.mlp(feat)
        if self.conv_last is not None:
            return feat, self.conv_last(feat)
        return feat
