"""Algorithmic HBM bytes AND floating-point operations of the hot-path operators.

Bytes: what each call must read and write if every input is read once and every output written once -- intermediates a
fused implementation keeps on chip count nothing.  Flops: the arithmetic of the reference's formulation (models/pointconv.py
:46-58, models/pwc3d_core.py:81-115, models/csrc/wrapper.py:47-52), multiply and add counted separately.  bench.py prices
every category at its roofline floor max(bytes / 8 TB/s, flops / 157.3 TFLOP/s) -- the fp32 matrix peak of
MI355X_MICROARCH.md, which SURVEY 8(d) names for KNN as well -- and divides by the measured duration; DESIGN.md section 5
states the same formulas.  All tensors fp32 (4 B), indices int64 (8 B)."""

HBM_PEAK_BYTES_PER_S = 8.0e12      # MI355X_MICROARCH.md: HBM3E
MFMA_F32_PEAK_FLOPS = 157.3e12     # MI355X_MICROARCH.md: dense fp32 matrix peak

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


# ------------------------------------------------------------------ flops (right-hand sides)
def knn_flops(B, Q, M, D):
    """B * Q * M pair evaluations of 2 D + 3 flops: the D-term dot product (2 D - 1), the -2 scaling and the two norm adds
    (wrapper.py:47-52); selection is not arithmetic and counts nothing."""
    return B * Q * M * (2 * D + 3)


def fps_flops(B, N, S):
    return B * S * N * 8   # three differences, three squares, two adds per (sample, point); latency-bound all the same


def pointconv_flops(B, Q, C, Cout, k=K):
    """pointconv.py:46-58 per query: the weighted sums 2 * 16 * k * (C + 3), nn.Linear 2 * 16 (C + 3) * Cout, the weight net
    MLP2d(3, [8, 16]) on k neighbours 2 k (3 * 8 + 8 * 16)."""
    return B * Q * (2 * 16 * k * (C + 3) + 2 * 16 * (C + 3) * Cout + 2 * k * (3 * 8 + 8 * 16))


def correlation3d_flops(B, N, C, k=K):
    """pwc3d_core.py:81-115: cost_mlp's two 1x1 layers over [B, 2C+3 -> C -> C, N, k], the two weight nets MLP2d(3, [8, 8, C])
    and the two weighted sums over k."""
    weight_net = 2 * (3 * 8 + 8 * 8 + 8 * C)
    return B * N * k * (2 * ((2 * C + 3) * C + C * C) + 2 * weight_net + 2 * 2 * C)


def conv1x1_flops(B, Cin, Cout, P):
    return 2 * B * P * Cin * Cout


def correlation2d_flops(B, C, H, W, md=4):
    """2 C flops per (pixel, displacement) pair that lies inside the image: sum over dy of (H - |dy|) times the same in x
    (= 2 C (9 H - 20)(9 W - 20) at md = 4)."""
    rows = sum(max(0, H - abs(d)) for d in range(-md, md + 1))
    cols = sum(max(0, W - abs(d)) for d in range(-md, md + 1))
    return 2 * B * C * rows * cols


def hotpath_flops(B, sizes, n_points=8192):
    """Per-step floating-point operations of every category of rpeflow_amd.hotpath.HotPathWorkload, same keys as
    hotpath_bytes.  The samplers, warps and gathers are a handful of flops per byte moved (8 per bilinear tap set, 2 per
    correlated channel): counted, never the bound."""
    N = [n_points, 4096, 2048, 1024, 512, 256]
    C, HW = PYRAMID_C, [h * w for h, w in sizes]
    out = {}
    out["fps+pyramid"] = fps_flops(2 * B, n_points, 4096)
    fp = 2 * conv1x1_flops(B, 3, C[0], N[0]) + conv1x1_flops(B, C[0], C[0], N[0])
    for i in range(5):
        fp += (conv1x1_flops(B, C[i], C[i], N[i]) + conv1x1_flops(B, C[i], C[i + 1], N[i]) + knn_flops(B, N[i + 1], N[i], 3)
               + pointconv_flops(B, N[i + 1], C[i + 1], C[i + 1]))
    out["feature_pyramid_3d"] = 2 * fp
    lv = range(1, 6)
    out["knn2d_k1"] = sum(2 * knn_flops(B, HW[l], N[l], 2) for l in lv)
    out["knn3d_k16"] = sum(knn_flops(B, N[l], N[l], 3) for l in lv)
    out["project_feat"] = sum(2 * B * HW[l] * (2 * C[l] + 81 + 64) for l in lv)                       # the per-pixel channel means
    out["grid_sample"] = sum(8 * B * N[l] * (2 * C[l] + 83 + EVENT_C[l] + 64) for l in lv)           # four taps, weights applied
    out["backwarp_2d"] = sum(8 * B * HW[l] * C[l] for l in range(1, 5))
    interp = lambda Cf, M, Q: knn_flops(B, Q, M, 3) + B * Q * (3 * 2 * Cf + 30)                       # k = 3: weights, normalisation, weighted sum
    out["knn_interpolation"] = sum(interp(67, N[l + 1], N[l]) for l in range(1, 5)) + sum(interp(3, N[i + 1], N[i]) for i in range(5))
    out["backwarp_3d"] = sum(interp(3, N[l], N[l]) + 6 * B * N[l] for l in range(1, 5))
    out["correlation3d"] = sum(knn_flops(B, N[l], N[l], 3) + correlation3d_flops(B, N[l], C[l]) for l in lv)
    out["correlation2d"] = sum(correlation2d_flops(B, C[l], *sizes[l]) for l in lv)
    out["flow_estimator_3d"] = sum(2 * conv1x1_flops(B, C[l], 64, N[l]) + pointconv_flops(B, N[l], 195, 128) + pointconv_flops(B, N[l], 128, 128)
                                   + conv1x1_flops(B, 128, 128, N[l]) + conv1x1_flops(B, 128, 64, N[l]) for l in lv)
    out["project_pc2image"] = sum(2 * B * N[l] * 4 for l in lv)
    out["flow_head_3d"] = sum(conv1x1_flops(B, 64, 3, N[l]) + 3 * B * N[l] for l in lv)
    out["torch_glue"] = out["project_pc2image"] + out["flow_head_3d"]
    return out


LATENCY_BOUND = {"fps+pyramid"}   # 4095 dependent iterations per cloud: neither roofline applies


def floor_seconds(nbytes, flops):
    """(roofline floor in seconds, which roofline sets it)."""
    t_hbm, t_mfma = nbytes / HBM_PEAK_BYTES_PER_S, flops / MFMA_F32_PEAK_FLOPS
    return (t_mfma, "mfma") if t_mfma > t_hbm else (t_hbm, "hbm")
