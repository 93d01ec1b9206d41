// Correlation3D (models/pwc3d_core.py:69-117) behind its neighbour search as TWO kernels for gfx950, one wave per point.
//
// corr3d_cost_kernel -- the point-to-neighbour cost (:84-98).  For point n with neighbours j < 16 in cloud 2:
//     hidden[j][c] = leaky(P1[n][c] + P2[idx_j][c] + Wc[c].rel_j)                    first cost_mlp layer; it is linear in
//                    the concatenation [feat1 | feat2_nbr | rel] (:92-94), so its feat1 / feat2 blocks are applied per
//                    POINT by the caller (P1 = Wa feat1 + b, P2 = Wb feat2, channel-last rows: one small GEMM) and only
//                    the gather, the 3-wide rel block and the activation run per pair -- [B,2C+3,N,16] never exists
//     cost[j][c']  = leaky(sum_c W2[c'][c] hidden[j][c] + b2[c'])                    second layer: the 16 neighbours of a point
//                    are the M = 16 rows of v_mfma_f32_16x16x4_f32 tiles; `hidden` is produced directly in the A-operand
//                    layout (lane = (k-slot, neighbour)), W2 comes pre-packed in B-fragment order
//     wn2[j][c']   = relu(W3 relu(W2n relu(W1n rel_j + b1) + b2) + b3)[c']           weight_net2 (:96): 3->8->8 on the
//                    VALU per neighbour, the last 8->C layer as two more MFMA steps per tile, same D layout as `cost`
//     p2n[n][c']   = sum_j wn2[j][c'] * cost[j][c']                                  (:98) in-lane + two cross-lane adds
//   written channel-last [B,N,Cp]: the second hop gathers whole rows of it.
// corr3d_n2n_kernel -- the neighbour-to-neighbour cost (:106-115): out[c][n] = sum_j wn1(rel1_j)[c] * p2n[idx1_j][c].
//
// Nothing of size [B,C,N,16] is written to memory (the two-kernel path before wrote two such tensors).  fp32 throughout.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Net3 {  // MLP2d(3, [8, 8, C], relu) (pwc3d_core.py:66-67): first two layers plain, last layer packed
    const float *w1, *b1, *w2, *b2;  // [8,3] [8] [8,8] [8]
    const f32x2 *w3p;                // [T][64] float2: lane (kk, n), element s = W3[16t + n][4s + kk]
    const float *b3;                 // [Cp]
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float leaky(float x, float slope) { return x >= 0.f ? x : x * slope; }

// relu(W2 relu(W1 rel + b1) + b2): the 3 -> 8 -> 8 front of a weight net for this lane's neighbour; returns the two values
// this lane feeds to the last layer's MFMA steps (k-slot kk: hidden units kk and 4 + kk).
__device__ __forceinline__ void net_front(const Net3 &net, const float (&rel)[3], int kk, float &a0, float &a1) {
    float h1[8], h2[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float s = net.b1[o];
#pragma unroll
        for (int d = 0; d < 3; ++d) s = __fmaf_rn(net.w1[o * 3 + d], rel[d], s);
        h1[o] = rpe_relu(s);
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float s = net.b2[o];
#pragma unroll
        for (int i = 0; i < 8; ++i) s = __fmaf_rn(net.w2[o * 8 + i], h1[i], s);
        h2[o] = rpe_relu(s);
    }
    a0 = kk == 0 ? h2[0] : kk == 1 ? h2[1] : kk == 2 ? h2[2] : h2[3];
    a1 = kk == 0 ? h2[4] : kk == 1 ? h2[5] : kk == 2 ? h2[6] : h2[7];
}

// sum over the 16 rows (neighbours) of a D tile: in-lane over the 4 registers, then lanes l, l^16, l^32, l^48
__device__ __forceinline__ float column_sum(f32x4 v) {
    float s = (v[0] + v[1]) + (v[2] + v[3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    return s;
}

// SPLIT (wide layers): the four waves of a workgroup share ONE point, wave w takes output tiles w, w + 4, ... -- a quarter
// of the MFMA chain and of the weight fragments each (one wave per point streamed T^2 KB of fragments through a chain of
// 4 T^2 MFMAs: 45 us at C = 192 whatever the point count); the gather and the first layer are repeated per wave, they
// are 1/T of the work.  Else four points a workgroup, one wave each.
template <int T, bool SPLIT>  // Cp = 16 * T channels
__global__ __launch_bounds__(256) void corr3d_cost_kernel(
    const float *__restrict__ p1rows, const float *__restrict__ p2rows, const float *__restrict__ wc4, const f32x4 *__restrict__ w2p,
    const float *__restrict__ b2, Net3 net, const float *__restrict__ xyz_q, int64_t q_sb, int64_t q_sd, int64_t q_sn,
    const float *__restrict__ xyz_s, int64_t s_sb, int64_t s_sd, int64_t s_sn, const int64_t *__restrict__ knn, int64_t knn_sq, int N, int M,
    float slope, float *__restrict__ p2n_rows) {
    constexpr int Cp = 16 * T;
    __shared__ f32x4 wc_lds[Cp];
    for (int i = threadIdx.x; i < Cp; i += 256) wc_lds[i] = ((const f32x4 *)wc4)[i];
    __syncthreads();
    const int lane = rpe_lane(), kk = lane >> 4, j = lane & 15;
    const int wave = rpe_uniform((int)(threadIdx.x >> 6));
    const int n = SPLIT ? (int)blockIdx.x : blockIdx.x * 4 + wave;
    constexpr int TP = SPLIT ? (T + 3) / 4 : T;  // output tiles of this wave: t = first + step * p
    const int first = SPLIT ? wave : 0;
    constexpr int step = SPLIT ? 4 : 1;
    const int b = blockIdx.y;
    if (n >= N) return;
    const int idx = (int)knn[((int64_t)b * N + n) * knn_sq + j];
    float rel[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        rel[d] = xyz_s[(int64_t)b * s_sb + d * s_sd + (int64_t)idx * s_sn] - xyz_q[(int64_t)b * q_sb + d * q_sd + (int64_t)n * q_sn];
    float na0, na1;
    net_front(net, rel, kk, na0, na1);

    const f32x4 *p1 = (const f32x4 *)(p1rows + ((int64_t)b * N + n) * Cp) + kk;
    const f32x4 *p2 = (const f32x4 *)(p2rows + ((int64_t)b * M + idx) * Cp) + kk;
    f32x4 acc[TP];
#pragma unroll
    for (int p = 0; p < TP; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the row pieces and the weight fragments of K-group g + 1 are requested before the MFMAs of group g
    auto load_w = [&](int g, f32x4 (&bf)[TP]) {
        const f32x4 *wf = w2p + (int64_t)min(g, T - 1) * T * 64 + lane;
#pragma unroll
        for (int p = 0; p < TP; ++p) bf[p] = wf[min(first + step * p, T - 1) * 64];
    };
    f32x4 v1 = p1[0], v2 = p2[0];
    f32x4 bfa[TP], bfb[TP];
    load_w(0, bfa);
#pragma unroll 2
    for (int g = 0; g < T; ++g) {
        const f32x4 a1 = v1, a2 = v2;
        if (g + 1 < T) v1 = p1[4 * (g + 1)], v2 = p2[4 * (g + 1)];
        f32x4 (&cur)[TP] = (g & 1) ? bfb : bfa;
        f32x4 (&nxt)[TP] = (g & 1) ? bfa : bfb;
        if (g + 1 < T) load_w(g + 1, nxt);
        float hid[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 w = wc_lds[16 * g + 4 * kk + s];
            float v = a1[s] + a2[s];
            v = v + (w[0] * rel[0] + w[1] * rel[1] + w[2] * rel[2]);
            hid[s] = leaky(v, slope);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < TP; ++p) acc[p] = mfma16(hid[s], cur[p][s], acc[p]);
    }
    float *o = p2n_rows + ((int64_t)b * N + n) * Cp;
#pragma unroll
    for (int p = 0; p < TP; ++p) {
        const int t = first + step * p;
        if (t >= T) continue;  // (wave-uniform)
        const int c = 16 * t + j;  // D layout: column = lane & 15
        const float bias2 = b2[c], bias3 = net.b3[c];
        const f32x2 w3 = net.w3p[t * 64 + lane];
        f32x4 wn = f32x4{0.f, 0.f, 0.f, 0.f};
        wn = mfma16(na0, w3[0], wn);
        wn = mfma16(na1, w3[1], wn);
        f32x4 prod;
#pragma unroll
        for (int r = 0; r < 4; ++r) prod[r] = rpe_relu(wn[r] + bias3) * leaky(acc[p][r] + bias2, slope);
        const float s = column_sum(prod);
        if (kk == 0) o[c] = s;
    }
}

template <int T>
__global__ __launch_bounds__(256) void corr3d_n2n_kernel(const float *__restrict__ p2n_rows, Net3 net, const float *__restrict__ xyz, int64_t x_sb,
                                                         int64_t x_sd, int64_t x_sn, const int64_t *__restrict__ knn, int64_t knn_sq, int N,
                                                         int C, float *__restrict__ out) {
    constexpr int Cp = 16 * T;
    const int lane = rpe_lane(), kk = lane >> 4, j = lane & 15;
    const int n = blockIdx.x * 4 + rpe_uniform((int)(threadIdx.x >> 6));
    const int b = blockIdx.y;
    if (n >= N) return;
    const int64_t *row = knn + ((int64_t)b * N + n) * knn_sq;
    const int idx = (int)row[j];
    float rel[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        rel[d] = xyz[(int64_t)b * x_sb + d * x_sd + (int64_t)idx * x_sn] - xyz[(int64_t)b * x_sb + d * x_sd + (int64_t)n * x_sn];
    float na0, na1;
    net_front(net, rel, kk, na0, na1);
    const float *src[4];  // D layout: register r <-> neighbour 4*kk + r
#pragma unroll
    for (int r = 0; r < 4; ++r) src[r] = p2n_rows + ((int64_t)b * N + (int)row[4 * kk + r]) * Cp + j;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int c = 16 * t + j;
        const f32x2 w3 = net.w3p[t * 64 + lane];
        f32x4 wn = f32x4{0.f, 0.f, 0.f, 0.f};
        wn = mfma16(na0, w3[0], wn);
        wn = mfma16(na1, w3[1], wn);
        const float bias3 = net.b3[c];
        f32x4 prod;
#pragma unroll
        for (int r = 0; r < 4; ++r) prod[r] = rpe_relu(wn[r] + bias3) * src[r][16 * t];
        const float s = column_sum(prod);
        if (kk == 0 && c < C) out[((int64_t)b * C + c) * N + n] = s;
    }
}

template <int T>
int launch_cost(const float *p1rows, const float *p2rows, const float *wc4, const float *w2p, const float *b2, const Net3 &net, const float *xyz_q,
                int64_t q_sb, int64_t q_sd, int64_t q_sn, const float *xyz_s, int64_t s_sb, int64_t s_sd, int64_t s_sn, const int64_t *knn,
                int64_t knn_sq, int B, int N, int M, float slope, float *p2n_rows, hipStream_t st) {
    constexpr bool kSplit = T >= 12;  // (measured: 46 -> 24 us at C = 192; C = 128 level, C <= 96 slower: the weight net and gather repeat per wave)
    hipLaunchKernelGGL((corr3d_cost_kernel<T, kSplit>), dim3(kSplit ? N : (N + 3) / 4, B), dim3(256), 0, st, p1rows, p2rows, wc4, (const f32x4 *)w2p, b2, net, xyz_q, q_sb,
                       q_sd, q_sn, xyz_s, s_sb, s_sd, s_sn, knn, knn_sq, N, M, slope, p2n_rows);
    return rpe_launch_status();
}

template <int T>
int launch_n2n(const float *p2n_rows, const Net3 &net, const float *xyz, int64_t x_sb, int64_t x_sd, int64_t x_sn, const int64_t *knn, int64_t knn_sq,
               int B, int N, int C, float *out, hipStream_t st) {
    hipLaunchKernelGGL(corr3d_n2n_kernel<T>, dim3((N + 3) / 4, B), dim3(256), 0, st, p2n_rows, net, xyz, x_sb, x_sd, x_sn, knn, knn_sq, N, C, out);
    return rpe_launch_status();
}

}  // namespace

#define RPE_C3_DISPATCH(T, CALL)    \
    switch (T) {                    \
        case 1: return CALL(1);     \
        case 2: return CALL(2);     \
        case 4: return CALL(4);     \
        case 6: return CALL(6);     \
        case 8: return CALL(8);     \
        case 12: return CALL(12);   \
        default: return RPE_EUNSUPPORTED; \
    }

RPE_API int rpe_corr3d_cost(const float *p1_rows, const float *p2_rows, const float *wc4, const float *w2_packed, const float *b2,
                            const float *n_w1, const float *n_b1, const float *n_w2, const float *n_b2, const float *n_w3_packed,
                            const float *n_b3, const float *xyz_q, int64_t q_sb, int64_t q_sd, int64_t q_sn, const float *xyz_s,
                            int64_t s_sb, int64_t s_sd, int64_t s_sn, const int64_t *knn, int64_t knn_row_stride, int B, int Cp, int N,
                            int M, float leaky_slope, float *p2n_rows, rpe_stream_t stream) {
    if (!p1_rows || !p2_rows || !wc4 || !w2_packed || !b2 || !n_w1 || !n_b1 || !n_w2 || !n_b2 || !n_w3_packed || !n_b3 || !xyz_q || !xyz_s ||
        !knn || !p2n_rows)
        return RPE_EINVAL;
    if (B < 0 || N < 1 || M < 1 || Cp < 16 || (Cp & 15) || knn_row_stride < 16) return RPE_EINVAL;
    if (B == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    Net3 net{n_w1, n_b1, n_w2, n_b2, (const f32x2 *)n_w3_packed, n_b3};
    hipStream_t st = (hipStream_t)stream;
#define RPE_C3_COST(T) launch_cost<T>(p1_rows, p2_rows, wc4, w2_packed, b2, net, xyz_q, q_sb, q_sd, q_sn, xyz_s, s_sb, s_sd, s_sn, knn, knn_row_stride, B, N, M, leaky_slope, p2n_rows, st)
    RPE_C3_DISPATCH(Cp / 16, RPE_C3_COST)
#undef RPE_C3_COST
}

RPE_API int rpe_corr3d_n2n(const float *p2n_rows, const float *n_w1, const float *n_b1, const float *n_w2, const float *n_b2,
                           const float *n_w3_packed, const float *n_b3, const float *xyz, int64_t x_sb, int64_t x_sd, int64_t x_sn,
                           const int64_t *knn, int64_t knn_row_stride, int B, int C, int Cp, int N, float *out, rpe_stream_t stream) {
    if (!p2n_rows || !n_w1 || !n_b1 || !n_w2 || !n_b2 || !n_w3_packed || !n_b3 || !xyz || !knn || !out) return RPE_EINVAL;
    if (B < 0 || N < 1 || C < 1 || Cp < C || (Cp & 15) || knn_row_stride < 16) return RPE_EINVAL;
    if (B == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    Net3 net{n_w1, n_b1, n_w2, n_b2, (const f32x2 *)n_w3_packed, n_b3};
    hipStream_t st = (hipStream_t)stream;
#define RPE_C3_N2N(T) launch_n2n<T>(p2n_rows, net, xyz, x_sb, x_sd, x_sn, knn, knn_row_stride, B, N, C, out, st)
    RPE_C3_DISPATCH(Cp / 16, RPE_C3_N2N)
#undef RPE_C3_N2N
}
