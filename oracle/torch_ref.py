"""PyTorch-CPU port of the reference's fallback path -- TEST INFRASTRUCTURE, NOT PRODUCT.

What this is for: (1) bench.py's ``cpu_baseline`` leg times THIS module on the
GPU box's host cores ("kind": "port"); (2) tests use it as a second, op-for-op
checker next to the explicit-rounding oracle in oracle.py.  Only tests/,
bench.py's cpu_baseline leg and __graft_entry__.smoke() may import it.

Every function issues the same sequence of torch ops as the reference function
it cites (paths relative to /root/reference), so its cost on a CPU is the
reference's cost and -- on the same host and torch build -- its results are the
reference's results (tests/test_torch_ref_golden.py pins that against
tests/golden/).  Across hosts MKL may round matmul differently;
bit-level judgements therefore use oracle.py, not this file.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------- models/csrc/wrapper.py
def squared_distance(xyz1, xyz2):
    """wrapper.py:40-52: -2 q.p^T, then += |q|^2, then += |p|^2, in place."""
    assert xyz1.shape[-1] == xyz2.shape[-1] and xyz1.shape[-1] <= 3
    b, n1, n2 = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
    d = -2 * torch.matmul(xyz1, xyz2.transpose(1, 2))
    d += (xyz1 ** 2).sum(-1).view(b, n1, 1)
    d += (xyz2 ** 2).sum(-1).view(b, 1, n2)
    return d


def k_nearest_neighbor(input_xyz, query_xyz, k, cpp_impl=True):
    """wrapper.py:106-127, CPU branch: full distance matrix + topk."""
    if input_xyz.shape[1] <= 3:
        assert query_xyz.shape[1] == input_xyz.shape[1]
        input_xyz = input_xyz.transpose(1, 2).contiguous()
        query_xyz = query_xyz.transpose(1, 2).contiguous()
    return squared_distance(query_xyz, input_xyz).topk(k, dim=2, largest=False).indices.to(torch.long)


def furthest_point_sampling(xyz, n_samples, cpp_impl=True):
    """wrapper.py:75-103, CPU branch: n_samples iterations of ~6 small tensor ops."""
    assert xyz.shape[2] == 3 and xyz.shape[1] > n_samples
    b, n, _ = xyz.shape
    picked = torch.zeros(b, n_samples, dtype=torch.int64, device=xyz.device)
    dist = torch.full((b, n), 1e10, device=xyz.device)
    rows = torch.arange(b, device=xyz.device)
    cur = torch.zeros(b, dtype=torch.int64, device=xyz.device)
    for i in range(n_samples):
        picked[:, i] = cur
        centre = xyz[rows, cur, :].view(b, 1, 3)
        nd = ((xyz - centre) ** 2).sum(-1)
        closer = nd < dist
        dist[closer] = nd[closer]
        cur = dist.max(-1)[1]
    return picked


def correlation2d(input1, input2, max_displacement, cpp_impl=True):
    """wrapper.py:56-65: 81 shifted products, each reduced with mean over channels."""
    h, w = input1.shape[2:]
    md = max_displacement
    padded = F.pad(input2, [md] * 4)
    planes = []
    for i in range(2 * md + 1):
        for j in range(2 * md + 1):
            planes.append((input1 * padded[:, :, i:i + h, j:j + w]).mean(1, keepdim=True))
    return torch.cat(planes, 1)


# ---------------------------------------------------------------- models/utils.py
def batch_indexing_channel_first(data, indices):
    """utils.py:119-137: torch.gather with the index expanded over channels."""
    b, c = data.shape[:2]
    shape = list(indices.shape[1:])
    flat = indices.reshape(b, 1, -1).expand(b, c, -1)
    return torch.gather(data, 2, flat.to(torch.int64)).view([b, c] + shape)


def batch_indexing_channel_last(data, indices):
    """utils.py:101-116: advanced indexing with a broadcast batch index."""
    b = data.shape[0]
    bidx = torch.arange(b, device=indices.device).view([b] + [1] * (indices.dim() - 1)).expand(indices.shape)
    return data[bidx, indices.to(torch.long)] if data.dim() == 2 else data[bidx, indices.to(torch.long), :]


def knn_interpolation(input_xyz, input_features, query_xyz, k=3):
    """utils.py:140-156."""
    knn = k_nearest_neighbor(input_xyz, query_xyz, k)
    nbr = batch_indexing_channel_first(input_xyz, knn)
    dist = torch.linalg.norm(nbr - query_xyz[..., None], dim=1).clamp(1e-8)
    w = 1.0 / dist
    w = w / w.sum(-1, keepdim=True)
    feats = batch_indexing_channel_first(input_features, knn)
    return (feats * w[:, None]).sum(-1)


def backwarp_3d(xyz1, xyz2, flow12, k=3):
    """utils.py:159-169."""
    return xyz2 + knn_interpolation(xyz1 + flow12, -flow12, xyz2, k)


def _pixel_grid(b, h, w, device=None):
    xs = torch.arange(w, dtype=torch.float32, device=device)[None, None, :].expand(b, h, w)
    ys = torch.arange(h, dtype=torch.float32, device=device)[None, :, None].expand(b, h, w)
    return torch.stack([xs, ys], 1)


def backwarp_2d(x, flow12, padding_mode):
    """utils.py:186-198: normalise (pixel + flow) to [-1,1], F.grid_sample."""
    b, _, h, w = x.shape
    g = _pixel_grid(b, h, w, flow12.device) + flow12
    gn = torch.zeros_like(g)
    gn[:, 0] = 2.0 * g[:, 0] / (w - 1) - 1.0
    gn[:, 1] = 2.0 * g[:, 1] / (h - 1) - 1.0
    return F.grid_sample(x, gn.permute(0, 2, 3, 1), padding_mode=padding_mode, align_corners=True)


def grid_sample_wrapper(feat_2d, xy):
    """utils.py:288-294 (padding stays 'zeros')."""
    h, w = feat_2d.shape[2:]
    nx = 2.0 * xy[:, 0] / (w - 1) - 1.0
    ny = 2.0 * xy[:, 1] / (h - 1) - 1.0
    g = torch.cat([nx[:, :, None, None], ny[:, :, None, None]], -1)
    return F.grid_sample(feat_2d, g, "bilinear", align_corners=True)[..., 0]


@torch.no_grad()
def project_feat_with_nn_corr(xy, feat_2d, feat_3d, nn_indices=None):
    """utils.py:297-317."""
    b, _, h, w = feat_2d.shape
    grid = _pixel_grid(b, h, w, xy.device).reshape(b, 2, -1)
    if nn_indices is None:
        nn_indices = k_nearest_neighbor(xy, grid, k=1)[..., 0]
    f2 = batch_indexing_channel_first(grid_sample_wrapper(feat_2d, xy), nn_indices)
    f3 = batch_indexing_channel_first(feat_3d, nn_indices)
    off = batch_indexing_channel_first(xy, nn_indices) - grid
    corr = (f2 * feat_2d.reshape(b, -1, h * w)).mean(1, keepdim=True)
    return torch.cat([off, corr, f3], 1).reshape(b, -1, h, w)


class _Block(nn.Module):
    """Conv{1,2}dNormRelu (utils.py:7-62), norm None or batch_norm."""

    def __init__(self, cin, cout, dims, norm, activation):
        super().__init__()
        self.conv_fn = (nn.Conv1d if dims == 1 else nn.Conv2d)(cin, cout, 1)
        bn = nn.BatchNorm1d if dims == 1 else nn.BatchNorm2d
        self.norm_fn = bn(cout) if norm == "batch_norm" else nn.Identity()
        self.relu_fn = {"relu": nn.ReLU(), "leaky_relu": nn.LeakyReLU(0.1), None: nn.Identity()}[activation]

    def forward(self, x):
        return self.relu_fn(self.norm_fn(self.conv_fn(x)))


class _MLP(nn.Module):
    def __init__(self, cin, widths, dims, norm=None, activation="leaky_relu"):
        super().__init__()
        chans = [cin] + widths
        self.convs = nn.ModuleList(_Block(a, b, dims, norm, activation) for a, b in zip(chans[:-1], chans[1:]))

    def forward(self, x):
        for c in self.convs:
            x = c(x)
        return x


def MLP1d(cin, widths, norm=None, activation="leaky_relu"):
    return _MLP(cin, widths, 1, norm, activation)


def MLP2d(cin, widths, norm=None, activation="leaky_relu"):
    return _MLP(cin, widths, 2, norm, activation)


# ---------------------------------------------------------------- models/pointconv.py
class _PointConv(nn.Module):
    def __init__(self, in_channels, out_channels, norm=None, activation="leaky_relu", k=16):
        super().__init__()
        self.k = k
        self.weight_net = MLP2d(3, [8, 16], activation=activation)
        self.linear = nn.Linear(16 * (in_channels + 3), out_channels)
        self.norm_fn = nn.BatchNorm1d(out_channels) if norm == "batch_norm" else nn.Identity()
        self.activation_fn = nn.LeakyReLU(0.1)

    def _conv(self, xyz, features, centres, knn):
        """pointconv.py:43-59 / 98-120: every intermediate is a real tensor."""
        b, q = xyz.shape[0], centres.shape[2]
        rows = torch.cat([xyz, features], 1).transpose(1, 2)
        rel = batch_indexing_channel_first(xyz, knn) - centres[:, :, :, None]
        wts = self.weight_net(rel).transpose(1, 2)
        grouped = batch_indexing_channel_last(rows, knn)
        mixed = torch.matmul(wts, grouped).view(b, q, -1)
        return self.activation_fn(self.norm_fn(self.linear(mixed).float().transpose(1, 2)))


class PointConvDownSampling(_PointConv):
    def forward(self, xyz, features, sampled_xyz):
        return self._conv(xyz, features, sampled_xyz, k_nearest_neighbor(xyz, sampled_xyz, self.k))


class PointConvNoSampling(_PointConv):
    def forward(self, xyz, features, knn_indices=None):
        knn = knn_indices[:, :, :self.k] if knn_indices is not None else k_nearest_neighbor(xyz, xyz, self.k)
        return self._conv(xyz, features, xyz, knn)


# ---------------------------------------------------------------- models/pwc3d_core.py
def build_pc_pyramid(pc1, pc2, n_samples_list):
    """pwc3d_core.py:8-28."""
    b, _, n = pc1.shape
    both = furthest_point_sampling(torch.cat([pc1, pc2], 0).transpose(1, 2), max(n_samples_list))
    s1, s2 = both[:b], both[b:]
    lv0 = torch.arange(n, device=pc1.device)[None, :].expand(b, n)
    xyzs1, xyzs2, idx1, idx2 = [pc1], [pc2], [lv0], [lv0]
    for m in n_samples_list:
        idx1.append(s1[:, :m])
        idx2.append(s2[:, :m])
        xyzs1.append(batch_indexing_channel_first(pc1, s1[:, :m]))
        xyzs2.append(batch_indexing_channel_first(pc2, s2[:, :m]))
    return xyzs1, xyzs2, idx1, idx2


class FeaturePyramid3D(nn.Module):
    """pwc3d_core.py:31-57."""

    def __init__(self, n_channels, norm=None, k=16):
        super().__init__()
        self.level0_mlp = MLP1d(3, [n_channels[0], n_channels[0]])
        self.pyramid_mlps = nn.ModuleList(MLP1d(a, [a, b]) for a, b in zip(n_channels[:-1], n_channels[1:]))
        self.pyramid_convs = nn.ModuleList(PointConvDownSampling(b, b, norm=norm, k=k) for b in n_channels[1:])

    def forward(self, xyzs):
        feats = [self.level0_mlp(torch.zeros_like(xyzs[0]))]
        for i in range(len(xyzs) - 1):
            feats.append(self.pyramid_convs[i](xyzs[i], self.pyramid_mlps[i](feats[-1]), xyzs[i + 1]))
        return feats


class Correlation3D(nn.Module):
    """pwc3d_core.py:60-117, with the [B,2C+3,N,k] concatenation the reference builds."""

    def __init__(self, in_channels, out_channels, k=16):
        super().__init__()
        self.k = k
        self.cost_mlp = MLP2d(3 + 2 * in_channels, [out_channels, out_channels], activation="leaky_relu")
        self.weight_net1 = MLP2d(3, [8, 8, out_channels], activation="relu")
        self.weight_net2 = MLP2d(3, [8, 8, out_channels], activation="relu")

    def forward(self, xyz1, feat1, xyz2, feat2, knn_indices_1in1=None):
        b, c, n = feat1.shape
        knn12 = k_nearest_neighbor(input_xyz=xyz2, query_xyz=xyz1, k=self.k)
        rel2 = batch_indexing_channel_first(xyz2, knn12) - xyz1.view(b, 3, n, 1)
        nbr2 = batch_indexing_channel_first(feat2, knn12)
        stacked = torch.cat([feat1[:, :, :, None].expand(b, c, n, self.k), nbr2, rel2], 1)
        p2n = (self.weight_net2(rel2) * self.cost_mlp(stacked)).sum(3)
        knn11 = knn_indices_1in1 if knn_indices_1in1 is not None else k_nearest_neighbor(xyz1, xyz1, self.k)
        rel1 = batch_indexing_channel_first(xyz1, knn11) - xyz1.view(b, 3, n, 1)
        return (self.weight_net1(rel1) * batch_indexing_channel_first(p2n, knn11)).sum(3)


class FlowEstimator3D(nn.Module):
    """pwc3d_core.py:120-148."""

    def __init__(self, n_channels, norm=None, conv_last=True, k=16):
        super().__init__()
        self.point_conv1 = PointConvNoSampling(n_channels[0], n_channels[1], norm=norm, k=k)
        self.point_conv2 = PointConvNoSampling(n_channels[1], n_channels[2], norm=norm, k=k)
        self.mlp = MLP1d(n_channels[2], [n_channels[2], n_channels[3]])
        self.conv_last = nn.Conv1d(n_channels[3], 3, 1) if conv_last else None

    def forward(self, xyz, feat, knn_indices):
        feat = self.mlp(self.point_conv2(xyz, self.point_conv1(xyz, feat, knn_indices), knn_indices))
        return (feat, self.conv_last(feat)) if self.conv_last is not None else feat
