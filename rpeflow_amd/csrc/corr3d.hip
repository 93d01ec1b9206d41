// Correlation3D (models/pwc3d_core.py:69-117) -- the gather/weight-net/k-sum parts as two kernels.
//
// The reference builds [B,2C+3,N,k] by concatenation, runs cost_mlp on it, evaluates two
// 3->8->8->C weight nets as 1x1 convolutions over [B,*,N,k] tensors and reduces over k
// with separate multiply and sum kernels (~40 launches, each a pass over a tensor of
// B*C*N*16 floats).  Here:
//
//   corr3d_hidden_kernel   hidden[b,c,n,j] = leaky(P1[b,c,n] + P2[b,c,idx_j] + Wc[c,:].rel_j)
//                          where P1 = Wa.feat1 + bias and P2 = Wb.feat2 are per-POINT products the
//                          host computes with two small GEMMs: cost_mlp's first layer is linear in
//                          the concatenation [feat1 | feat2_nbr | rel] (pwc3d_core.py:92-94), so it
//                          splits by input block and only the second layer runs per (point, neighbour).
//   corr3d_wsum_kernel     out[b,c,n] = sum_j relu-MLP(rel_j)[c] * value[b,c,n,j]      (:96-98)
//                          or, with GATHER, value = p2n[b,c,idx_j]                     (:106-115)
//                          one thread per (point, neighbour); the 3->8->8 layers once per thread, the
//                          last layer and the product per channel, the k-sum on DPP row adds.
#include "common.h"

namespace {

constexpr int K = 16;

struct Net3 {  // MLP2d(3, [8, 8, C], relu), norm None (pwc3d_core.py:66-67)
    const float *w1, *b1, *w2, *b2, *w3, *b3;
};

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);  // out of row -> 0
    return v + __int_as_float(o);
}
// lane 15 of every 16-lane row ends up with the row's sum
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add<0x111>(v);
    v = dpp_add<0x112>(v);
    v = dpp_add<0x114>(v);
    v = dpp_add<0x118>(v);
    return v;
}

__global__ __launch_bounds__(256) void corr3d_hidden_kernel(
    const float *__restrict__ p1, const float *__restrict__ p2, const float *__restrict__ wc,
    const float *__restrict__ xyz_q, int64_t q_sb, int64_t q_sd, int64_t q_sn,
    const float *__restrict__ xyz_s, int64_t s_sb, int64_t s_sd, int64_t s_sn,
    const int64_t *__restrict__ knn, int64_t knn_sq, int C, int N, int M, float slope, float *__restrict__ hidden) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= N * K) return;
    const int n = p / K, j = p % K;
    const int64_t idx = knn[((int64_t)b * N + n) * knn_sq + j];
    float rel[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        rel[d] = xyz_s[(int64_t)b * s_sb + d * s_sd + idx * s_sn] - xyz_q[(int64_t)b * q_sb + d * q_sd + (int64_t)n * q_sn];
    const float *a1 = p1 + (int64_t)b * C * N + n;
    const float *a2 = p2 + (int64_t)b * C * M + idx;
    float *o = hidden + (int64_t)b * C * N * K + p;
    for (int c = 0; c < C; ++c) {
        float v = a1[(int64_t)c * N] + a2[(int64_t)c * M];
        v = v + (wc[c * 3] * rel[0] + wc[c * 3 + 1] * rel[1] + wc[c * 3 + 2] * rel[2]);
        o[(int64_t)c * N * K] = v >= 0.f ? v : v * slope;
    }
}

template <bool GATHER>
__global__ __launch_bounds__(256) void corr3d_wsum_kernel(
    const float *__restrict__ vals, Net3 net,
    const float *__restrict__ xyz_q, int64_t q_sb, int64_t q_sd, int64_t q_sn,
    const float *__restrict__ xyz_s, int64_t s_sb, int64_t s_sd, int64_t s_sn,
    const int64_t *__restrict__ knn, int64_t knn_sq, int C, int N, int M, float *__restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;  // N*K is a multiple of 16: rows never straddle points
    const int b = blockIdx.y;
    const bool live = p < N * K;
    const int n = live ? p / K : N - 1, j = p % K;
    const int64_t idx = knn[((int64_t)b * N + n) * knn_sq + j];
    float rel[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        rel[d] = xyz_s[(int64_t)b * s_sb + d * s_sd + idx * s_sn] - xyz_q[(int64_t)b * q_sb + d * q_sd + (int64_t)n * q_sn];
    float h1[8], h2[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float s = net.b1[o];
#pragma unroll
        for (int d = 0; d < 3; ++d) s = __fmaf_rn(net.w1[o * 3 + d], rel[d], s);
        h1[o] = fmaxf(s, 0.f);
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float s = net.b2[o];
#pragma unroll
        for (int i = 0; i < 8; ++i) s = __fmaf_rn(net.w2[o * 8 + i], h1[i], s);
        h2[o] = fmaxf(s, 0.f);
    }
    const float *v = GATHER ? vals + (int64_t)b * C * M + idx : vals + (int64_t)b * C * N * K + (int64_t)n * K + j;
    const int64_t vstride = GATHER ? (int64_t)M : (int64_t)N * K;
    float *o = out + (int64_t)b * C * N + n;
    for (int c = 0; c < C; ++c) {
        float w = net.b3[c];
#pragma unroll
        for (int i = 0; i < 8; ++i) w = __fmaf_rn(net.w3[c * 8 + i], h2[i], w);
        w = fmaxf(w, 0.f);
        const float t = row_sum16(w * v[(int64_t)c * vstride]);
        if (live && j == K - 1) o[(int64_t)c * N] = t;
    }
}

}  // namespace

RPE_API int rpe_corr3d_hidden(const float *p1, const float *p2, const float *wc, const float *xyz_q, int64_t q_sb,
                              int64_t q_sd, int64_t q_sn, const float *xyz_s, int64_t s_sb, int64_t s_sd, int64_t s_sn,
                              const int64_t *knn, int64_t knn_row_stride, int B, int C, int N, int M, float leaky_slope,
                              float *hidden, rpe_stream_t stream) {
    if (!p1 || !p2 || !wc || !xyz_q || !xyz_s || !knn || !hidden || B < 0 || C < 1 || N < 1 || M < 1) return RPE_EINVAL;
    if (knn_row_stride < K) return RPE_EINVAL;
    if (B == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    dim3 grid((N * K + 255) / 256, B);
    hipLaunchKernelGGL(corr3d_hidden_kernel, grid, dim3(256), 0, (hipStream_t)stream, p1, p2, wc, xyz_q, q_sb, q_sd, q_sn,
                       xyz_s, s_sb, s_sd, s_sn, knn, knn_row_stride, C, N, M, leaky_slope, hidden);
    return rpe_launch_status();
}

RPE_API int rpe_corr3d_weighted_sum(const float *vals, int gather, const float *w1, const float *b1, const float *w2,
                                    const float *b2, const float *w3, const float *b3, const float *xyz_q, int64_t q_sb,
                                    int64_t q_sd, int64_t q_sn, const float *xyz_s, int64_t s_sb, int64_t s_sd, int64_t s_sn,
                                    const int64_t *knn, int64_t knn_row_stride, int B, int C, int N, int M, float *out,
                                    rpe_stream_t stream) {
    if (!vals || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !xyz_q || !xyz_s || !knn || !out) return RPE_EINVAL;
    if (B < 0 || C < 1 || N < 1 || M < 1 || knn_row_stride < K) return RPE_EINVAL;
    if (B == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    Net3 net{w1, b1, w2, b2, w3, b3};
    dim3 grid((N * K + 255) / 256, B);
    hipStream_t st = (hipStream_t)stream;
    if (gather)
        hipLaunchKernelGGL(corr3d_wsum_kernel<true>, grid, dim3(256), 0, st, vals, net, xyz_q, q_sb, q_sd, q_sn, xyz_s, s_sb,
                           s_sd, s_sn, knn, knn_row_stride, C, N, M, out);
    else
        hipLaunchKernelGGL(corr3d_wsum_kernel<false>, grid, dim3(256), 0, st, vals, net, xyz_q, q_sb, q_sd, q_sn, xyz_s, s_sb,
                           s_sd, s_sn, knn, knn_row_stride, C, N, M, out);
    return rpe_launch_status();
}
