"""Algorithmic HBM bytes of the hot-path operators: what each call must read and write if every
input is read once and every output written once -- intermediates a fused implementation keeps on chip count nothing.
bench.py divides these by measured durations; DESIGN.md section 5 states the same formulas.  All tensors fp32 (4 B),
indices int64 (8 B)."""

PYRAMID_C = [16, 32, 64, 96, 128, 192]   # 2-D and 3-D pyramid widths (RPEFlow_core.py:174-177, 215-219)
EVENT_C = [32, 32, 64, 96, 128, 192]     # event pyramid widths (RPEFlow_core.py:181-184)
K = 16


def knn(B, Q, M, D, k):
    return 4 * B * D * (Q + M) + 8 * B * Q * k


def fps(B, N, S):
    return 12 * B * N + 8 * B * S


def gather(B, C, N, n_idx):
    return 4 * B * C * N + 8 * B * n_idx + 4 * B * C * n_idx


def correlation2d(B, C, H, W, md=4):
    return 2 * B * C * H * W * 4 + B * (2 * md + 1) ** 2 * H * W * 4


def backwarp_2d(B, C, HW):
    return 4 * B * HW * (2 * C + 2)


def project_feat(B, C2, C3, HW, N):
    return 4 * B * (C2 * HW + (C3 + 2) * N + (C3 + 3) * HW) + 8 * B * HW


def grid_sample(B, C, HW, N):
    return 4 * B * (C * HW + 2 * N + C * N)


def knn_interpolation(B, C, M, Q, k=3):
    return knn(B, Q, M, 3, k) + 4 * B * (C * M + C * Q)


def backwarp_3d(B, N, k=3):
    # utils.py:159-169: warp xyz1 by the flow, interpolate -flow from the warped cloud at xyz2 (KNN k=3), add to xyz2
    return knn_interpolation(B, 3, N, N, k) + 4 * B * 3 * N * 2


def pointconv(B, M, Q, C, Cout, with_knn):
    weights = 4 * (8 * 3 + 8 + 16 * 8 + 16 + 16 * (C + 3) * Cout + Cout)
    return (knn(B, Q, M, 3, K) if with_knn else 8 * B * Q * K) + 4 * B * (C + 3) * M + weights + 4 * B * Cout * Q


def correlation3d(B, N, C):
    weights = 4 * ((2 * C + 3) * C + C * C + 2 * (8 * 3 + 8 * 8 + 8 * C) + 4 * C)
    return knn(B, N, N, 3, K) + 4 * B * N * (2 * C + 6) + 8 * B * N * K + weights + 4 * B * C * N


def conv1x1(B, Cin, Cout, P):
    return 4 * B * P * (Cin + Cout) + 4 * Cin * Cout


def hotpath_bytes(B, sizes, n_points=8192):
    """Per-step algorithmic bytes of every category of rpeflow_amd.hotpath.HotPathWorkload (``sizes`` = its per-level
    (H, W)); category names are its span names."""
    N = [n_points, 4096, 2048, 1024, 512, 256]
    C, HW = PYRAMID_C, [h * w for h, w in sizes]
    out = {}
    out["fps+pyramid"] = fps(2 * B, n_points, 4096) + 2 * gather(B, 3, n_points, 4096)
    fp = 2 * conv1x1(B, 3, C[0], N[0]) + conv1x1(B, C[0], C[0], N[0])
    for i in range(5):
        fp += conv1x1(B, C[i], C[i], N[i]) + conv1x1(B, C[i], C[i + 1], N[i]) + pointconv(B, N[i], N[i + 1], C[i + 1], C[i + 1], True)
    out["feature_pyramid_3d"] = 2 * fp  # both clouds
    lv = range(1, 6)
    out["knn2d_k1"] = sum(2 * knn(B, HW[l], N[l], 2, 1) for l in lv)
    out["knn3d_k16"] = sum(knn(B, N[l], N[l], 3, K) for l in lv)
    out["project_feat"] = sum(2 * project_feat(B, C[l], C[l], HW[l], N[l]) + project_feat(B, 81, C[l] + 2, HW[l], N[l])
                              + project_feat(B, 64, 64, HW[l], N[l]) for l in lv)
    out["grid_sample"] = sum(2 * grid_sample(B, C[l], HW[l], N[l]) + grid_sample(B, 83, HW[l], N[l])
                             + grid_sample(B, EVENT_C[l], HW[l], N[l]) + grid_sample(B, 64, HW[l], N[l]) for l in lv)
    out["backwarp_2d"] = sum(backwarp_2d(B, C[l], HW[l]) for l in range(1, 5))
    out["knn_interpolation"] = (sum(knn_interpolation(B, 67, N[l + 1], N[l]) for l in range(1, 5))
                                + sum(knn_interpolation(B, 3, N[i + 1], N[i]) for i in range(5)))
    out["backwarp_3d"] = sum(backwarp_3d(B, N[l]) for l in range(1, 5))
    out["correlation3d"] = sum(correlation3d(B, N[l], C[l]) for l in lv)
    out["correlation2d"] = sum(correlation2d(B, C[l], *sizes[l]) for l in lv)
    out["flow_estimator_3d"] = sum(2 * conv1x1(B, C[l], 64, N[l]) + pointconv(B, N[l], N[l], 195, 128, False)
                                   + pointconv(B, N[l], N[l], 128, 128, False) + conv1x1(B, 128, 128, N[l])
                                   + conv1x1(B, 128, 64, N[l]) for l in lv)
    # project_pc2image + rescale of both clouds (read x, y; write u, v), and the flow head (read 64 features + 3 flow, write 3);
    # a port of the reference passed as ``ops`` reports both, with its zero fills, under "torch_glue" -- the same bytes
    out["project_pc2image"] = sum(4 * 2 * B * N[l] * (2 + 2) for l in lv)
    out["flow_head_3d"] = sum(4 * B * N[l] * (64 + 3 + 3) + 4 * 64 * 3 for l in lv)
    out["torch_glue"] = out["project_pc2image"] + out["flow_head_3d"]
    return out
