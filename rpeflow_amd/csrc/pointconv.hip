// PointConv grouping for gfx950: KNN gather + weight-net MLP + k-reduction in one kernel.
//
// Covers the part of PointConvDownSampling.forward / PointConvNoSampling.forward between
// the KNN and the nn.Linear (models/pointconv.py:48-57, 107-118):
//   rel_j   = xyz[:, idx_j] - q_xyz                          (:50-51)
//   wn_j    = leaky(W2 leaky(W1 rel_j + b1) + b2)  in R^16   (weight_net, MLP2d 3->8->16, :12,52)
//   out[q][w*(C+3)+c] = sum_j wn_j[w] * feats_cl[idx_j][c]   (matmul + view, :54-57)
// The reference materialises knn_xyz, weights, knn_features ([B,Q,16,C+3]: 208 MB at
// level 1) and the product as separate tensors; here one wave owns one query point:
// lanes 0..15 evaluate the weight net for the 16 neighbours, the 16x16 weights go through
// 1 KB of LDS as broadcasts, and the lanes then walk the C+3 channels of each gathered
// (channel-last, contiguous) feature row, accumulating 16 outputs per channel in registers.
// No MFMA: per query this is a 16x16 by 16x(C+3) product on gathered rows.
#include "common.h"

namespace {

constexpr int K = 16;   // neighbours used (pointconv.py:9 default, pwc3d k=16)
constexpr int NW = 16;  // weight-net outputs
constexpr int WAVES = 4;

struct WeightNet {
    const float *w1, *b1, *w2, *b2;  // [8,3] [8] [16,8] [16], contiguous
};

__device__ __forceinline__ float leaky(float x, float slope) { return x >= 0.f ? x : x * slope; }

template <int R>  // channel rounds: C+3 <= 64*R
__global__ __launch_bounds__(WAVES * RPE_WAVE) void pointconv_group_kernel(
    const float *__restrict__ xyz, int64_t x_sb, int64_t x_sd, int64_t x_sn,
    const float *__restrict__ q_xyz, int64_t q_sb, int64_t q_sd, int64_t q_sn,
    const float *__restrict__ feats_cl, int CF, int M,
    const int64_t *__restrict__ knn, int64_t knn_sq, WeightNet wn, float slope, int Q, float *__restrict__ out) {
    __shared__ float wl[WAVES][K][NW];
    const int lane = rpe_lane();
    const int wave = rpe_uniform((int)(threadIdx.x >> 6));
    const int b = blockIdx.y;
    const int q = blockIdx.x * WAVES + wave;
    if (q >= Q) return;  // wave-uniform; no block-level barrier below

    // ---- weight net, lane j (duplicated in the other three 16-lane groups)
    const int j = lane & (K - 1);
    const int idx = (int)knn[((int64_t)b * Q + q) * knn_sq + j];
    const float *xb = xyz + (int64_t)b * x_sb, *qb = q_xyz + (int64_t)b * q_sb;
    float rel[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) rel[d] = xb[d * x_sd + (int64_t)idx * x_sn] - qb[d * q_sd + (int64_t)q * q_sn];
    float h[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float s = wn.b1[o];
#pragma unroll
        for (int d = 0; d < 3; ++d) s = __fmaf_rn(wn.w1[o * 3 + d], rel[d], s);
        h[o] = leaky(s, slope);
    }
    if (lane < K) {
#pragma unroll
        for (int o = 0; o < NW; ++o) {
            float s = wn.b2[o];
#pragma unroll
            for (int i = 0; i < 8; ++i) s = __fmaf_rn(wn.w2[o * 8 + i], h[i], s);
            wl[wave][j][o] = leaky(s, slope);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes are done
    __builtin_amdgcn_wave_barrier();

    // ---- weighted sum over the 16 gathered rows
    float acc[R][NW];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int w = 0; w < NW; ++w) acc[r][w] = 0.f;

    const float *fb = feats_cl + (int64_t)b * M * CF;
#pragma unroll 4
    for (int jj = 0; jj < K; ++jj) {
        const int src = rpe_readlane(idx, jj);
        const float *row = fb + (int64_t)src * CF;
        float wv[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) wv[w] = wl[wave][jj][w];  // broadcast reads
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int c = lane + r * RPE_WAVE;
            const float f = c < CF ? row[c] : 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) acc[r][w] = __fmaf_rn(wv[w], f, acc[r][w]);
        }
    }

    float *o = out + ((int64_t)b * Q + q) * (int64_t)NW * CF;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int c = lane + r * RPE_WAVE;
        if (c < CF) {
#pragma unroll
            for (int w = 0; w < NW; ++w) o[(int64_t)w * CF + c] = acc[r][w];
        }
    }
}

}  // namespace

RPE_API int rpe_pointconv_group(const float *xyz, int64_t x_sb, int64_t x_sd, int64_t x_sn, const float *q_xyz, int64_t q_sb,
                                int64_t q_sd, int64_t q_sn, const float *feats_cl, const int64_t *knn, int64_t knn_row_stride,
                                const float *w1, const float *b1, const float *w2, const float *b2, float leaky_slope, int B,
                                int M, int Q, int CF, float *out, rpe_stream_t stream) {
    if (!xyz || !q_xyz || !feats_cl || !knn || !w1 || !b1 || !w2 || !b2 || !out) return RPE_EINVAL;
    if (B < 0 || M < 1 || Q < 0 || CF < 1 || knn_row_stride < K) return RPE_EINVAL;
    if (CF > 4 * RPE_WAVE) return RPE_EUNSUPPORTED;
    if (B == 0 || Q == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    WeightNet wn{w1, b1, w2, b2};
    dim3 grid((Q + WAVES - 1) / WAVES, B), block(WAVES * RPE_WAVE);
    hipStream_t st = (hipStream_t)stream;
#define RPE_PC_LAUNCH(R)                                                                                                 \
    hipLaunchKernelGGL(pointconv_group_kernel<R>, grid, block, 0, st, xyz, x_sb, x_sd, x_sn, q_xyz, q_sb, q_sd, q_sn,     \
                       feats_cl, CF, M, knn, knn_row_stride, wn, leaky_slope, Q, out)
    if (CF <= 64) RPE_PC_LAUNCH(1);
    else if (CF <= 128) RPE_PC_LAUNCH(2);
    else if (CF <= 192) RPE_PC_LAUNCH(3);
    else RPE_PC_LAUNCH(4);
#undef RPE_PC_LAUNCH
    return rpe_launch_status();
}
