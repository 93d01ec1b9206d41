"""3-D pyramid blocks -- mirror of the reference's models/pwc3d_core.py.

Same classes, constructor arguments, forward signatures and state-dict keys.
KNN / FPS / gathers / PointConv grouping run as HIP kernels; the dense 1x1
convolutions stay on PyTorch (MIOpen / hipBLASLt).
"""
import torch
import torch.nn as nn

from .csrc import furthest_point_sampling, k_nearest_neighbor
from .pointconv import PointConvDownSampling, PointConvNoSampling
from .utils import MLP1d, MLP2d, batch_indexing_channel_first


def build_pc_pyramid(pc1, pc2, n_samples_list):
    """pwc3d_core.py:8-28: one FPS over both clouds, every level a prefix of its order."""
    batch_size, _, n_points = pc1.shape
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
        feats = [self.level0_mlp(torch.zeros_like(xyzs[0]))]
        for i in range(len(xyzs) - 1):
            feats.append(self.pyramid_convs[i](xyzs[i], self.pyramid_mlps[i](feats[-1]), xyzs[i + 1]))
        return feats


class Correlation3D(nn.Module):
    """pwc3d_core.py:60-117.

    The first cost_mlp layer is linear in the concatenation [feat1 | feat2_nbr | rel]
    (pwc3d_core.py:92-94), so it is applied per POINT before the gather --
    W_a feat1 + gather(W_b feat2) + W_c rel -- instead of per (point, neighbour) pair:
    k = 16 times fewer multiply-adds in that layer, and the [B, 2C+3, N, k] tensor is
    never built.  Results differ from the reference only by fp32 re-association."""

    def __init__(self, in_channels, out_channels, k=16):
        super().__init__()
        self.k = k
        self.cost_mlp = MLP2d(3 + 2 * in_channels, [out_channels, out_channels], activation="leaky_relu")
        self.weight_net1 = MLP2d(3, [8, 8, out_channels], activation="relu")
        self.weight_net2 = MLP2d(3, [8, 8, out_channels], activation="relu")

    def forward(self, xyz1, feat1, xyz2, feat2, knn_indices_1in1=None):
        batch_size, in_channels, n_points = feat1.shape
        knn_indices_1in2 = k_nearest_neighbor(input_xyz=xyz2, query_xyz=xyz1, k=self.k)
        knn_xyz2_norm = batch_indexing_channel_first(xyz2, knn_indices_1in2) - xyz1.view(batch_size, 3, n_points, 1)

        first = self.cost_mlp.convs[0]
        w = first.conv_fn.weight[:, :, 0, 0]  # [Cout, 2C+3], input order feat1 | feat2 | rel (:92)
        w_a, w_b, w_c = w[:, :in_channels], w[:, in_channels:2 * in_channels], w[:, 2 * in_channels:]
        part1 = torch.matmul(w_a, feat1) + first.conv_fn.bias[None, :, None]  # [B,Cout,N]
        part2 = batch_indexing_channel_first(torch.matmul(w_b, feat2), knn_indices_1in2)  # [B,Cout,N,k]
        part3 = torch.einsum("oc,bcnk->bonk", w_c, knn_xyz2_norm)
        hidden = first.relu_fn(first.norm_fn(part1[:, :, :, None] + part2 + part3))
        p2p_cost = self.cost_mlp.convs[1](hidden)

        weights2 = self.weight_net2(knn_xyz2_norm)
        p2n_cost = torch.sum(weights2 * p2p_cost, dim=3)

        if knn_indices_1in1 is not None:
            assert knn_indices_1in1.shape == torch.Size([batch_size, n_points, self.k])
        else:
            knn_indices_1in1 = k_nearest_neighbor(input_xyz=xyz1, query_xyz=xyz1, k=self.k)
        knn_xyz1_norm = batch_indexing_channel_first(xyz1, knn_indices_1in1) - xyz1.view(batch_size, 3, n_points, 1)
        weights1 = self.weight_net1(knn_xyz1_norm)
        n2n_cost = batch_indexing_channel_first(p2n_cost, knn_indices_1in1)
        return torch.sum(weights1 * n2n_cost, dim=3)


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
