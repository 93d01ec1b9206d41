// 1x1 convolution (Conv1d / Conv2d, stride 1) with its whole epilogue:
//   y[b][o][p] = act(scale[o] * sum_c W[o][c] x[b][c][p] + shift[o]) (+ residual[b][o][p]).
//
// The reference runs these as convolution + bias + BatchNorm + activation (+ a residual add): models/utils.py:7-62,
// models/restormer_arch.py:88-110, 169-222.  A library GEMM covers the sum only, so every such layer was GEMM + one
// element-wise pass (affine_act_kernel, 105 launches per forward) or GEMM + a copy of the residual in front of it (baddbmm).
// Most of these layers are latency-bound (8-15 us whatever the kernel does), so one launch less per layer is worth more than
// GEMM efficiency; measured on the whole forward that holds up to the largest ones (rpeflow_amd/utils.py, _PW_MAX_FLOPS).
//
// fp32 on v_mfma_f32_16x16x4_f32: A = a 16 x 4 piece of W (pre-packed on the host in fragment order: one coalesced 256-byte
// load per piece), B = 4 channels x 16 positions of x.  A lane loads FOUR consecutive positions of one channel row (one
// 16-byte load when the row stride allows) and feeds them to four position tiles, so the accumulators of a lane hold four
// consecutive positions of an output row and the epilogue stores 16 bytes per lane (positions p0 + 4 j + t, t = tile).
// A wave owns OT output tiles x 64 positions; the four waves of a workgroup take different output tiles of the same 64
// positions (their x loads hit in L1).  Summation order per output: channels ascending in groups of four -- fixed: the
// kernel is deterministic.
#include "common.h"

namespace {

typedef float pw_f32x4 __attribute__((ext_vector_type(4)));

struct PwArgs {
    const float *x;       // [B][Cin][P]
    const float *wpk;     // [n_otiles][ktiles][64]: lane (k, i) of piece (ot, kt) = W[16 ot + i][4 kt + k], zero outside
    const float *scale;   // [Cout] or NULL (1)
    const float *shift;   // [Cout] or NULL (0)
    const float *res;     // [B][Cout][P] or NULL
    float *y;             // [B][Cout][P]
    int Cin, Cout, ktiles, n_otiles, act;
    int64_t P, x_bstride, w_bstride;  // floats between samples of x; between per-sample weights (0: one weight for all)
    float slope;
    int64_t res_bstride;  // floats between samples of res (Cout * P for a dense tensor; larger for a channel slice of a wider one)
};

// Epilogue of pointwise_conv_kernel.  D layout: lane (g = lane >> 4, j), register r: output 16 ot + 4 g + r, position p0 + t of
// tile t.  v = act(scale * acc + shift) [+ residual].  The activation is ONE select per value whatever its kind -- identity,
// relu and leaky_relu are "u < 0 ? alt : u" with alt = u * 1, +0, u * slope (nn.ReLU / nn.LeakyReLU as ATen computes them on the
// reference's CPU path: NaN stays NaN, -0 stays -0) -- and the affine pairs are read a tile (four rows) at a time: with the
// tests on act / scale / shift inside the loops the compiler emitted ~4 scalar branches per output VALUE (260 branches a wave,
// a wait on every scale / shift pair).
template <int OT, bool VEC>
__device__ __forceinline__ void pw_finish(const PwArgs &a, const pw_f32x4 (&acc)[OT][4], int ot0, int g, int b, int64_t p0) {
    const int64_t P = a.P;
    const int oc0 = 16 * ot0 + 4 * g;
    const float alt_scale = a.act == 1 ? 0.0f : (a.act == 2 ? a.slope : 1.0f);
    const unsigned alt_keep = a.act == 1 ? 0u : ~0u;
    float *yb = a.y + ((int64_t)b * a.Cout + oc0) * P + p0;
    const float *rb = a.res ? a.res + (int64_t)b * a.res_bstride + (int64_t)oc0 * P + p0 : nullptr;
#pragma unroll
    for (int o = 0; o < OT; ++o) {
        // the tile's sums stay in the accumulation registers until its turn: the compiler would move all 16 OT vectors out
        // behind the loop (148 registers at four tiles a wave: two waves a SIMD; 119 and four waves this way)
        pw_f32x4 d[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            d[t] = acc[o][t];
            asm volatile("" : "+a"(d[t]));
        }
        float sc[4] = {1.0f, 1.0f, 1.0f, 1.0f}, sh[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // (the tile's affine pairs: one wait per tile)
        if (a.scale) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[r] = a.scale[min(oc0 + 16 * o + r, a.Cout - 1)];
        }
        if (a.shift) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sh[r] = a.shift[min(oc0 + 16 * o + r, a.Cout - 1)];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (oc0 + 16 * o + r >= a.Cout) continue;
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float u = sc[r] * d[t][r] + sh[r];
                const float alt = __uint_as_float(__float_as_uint(u * alt_scale) & alt_keep);
                v[t] = u < 0.f ? alt : u;
            }
            const int64_t off = (int64_t)(16 * o + r) * P;
            if (VEC) {  // P % 4 == 0 and 16-byte aligned bases: the four positions are in range together
                if (rb) {
                    const float4 rr = *reinterpret_cast<const float4 *>(rb + off);
                    v[0] += rr.x, v[1] += rr.y, v[2] += rr.z, v[3] += rr.w;
                }
                *reinterpret_cast<float4 *>(yb + off) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (p0 + t < P) yb[off + t] = rb ? v[t] + rb[off + t] : v[t];
            }
        }
    }
}

template <int OT, bool VEC>
__global__ __launch_bounds__(256) void pointwise_conv_kernel(PwArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = lane >> 4, j = lane & 15;
    const int ot0 = (blockIdx.y * 4 + wave) * OT;
    if (ot0 >= a.n_otiles) return;
    const int b = blockIdx.z;
    const int64_t P = a.P, p0 = (int64_t)blockIdx.x * 64 + 4 * j;
    const float *xb = a.x + (int64_t)b * a.x_bstride;
    const float *wpk = a.wpk + (int64_t)b * a.w_bstride;
    pw_f32x4 acc[OT][4];
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[o][t] = pw_f32x4{0.f, 0.f, 0.f, 0.f};
    const bool p_in = p0 < P;
    // four channel groups per round: all their loads are issued before the first matrix instruction waits for one
    constexpr int U = 4;  // (rounds of 2 / 8 / 12 groups measure the same: 174 / 161 / 177 against 165 us on the level-1 project_in)
    for (int kt0 = 0; kt0 < a.ktiles; kt0 += U) {
        float xv[U][4], av[U][OT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kt = kt0 + u, c = 4 * kt + k;
#pragma unroll
            for (int t = 0; t < 4; ++t) xv[u][t] = 0.f;
            if (c < a.Cin && p_in) {
                const float *row = xb + (int64_t)c * P + p0;
                if (VEC) {  // P % 4 == 0 and 16-byte aligned base: the four positions are in range together
                    const float4 v = *reinterpret_cast<const float4 *>(row);
                    xv[u][0] = v.x, xv[u][1] = v.y, xv[u][2] = v.z, xv[u][3] = v.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) xv[u][t] = p0 + t < P ? row[t] : 0.f;
                }
            }
#pragma unroll
            for (int o = 0; o < OT; ++o)
                av[u][o] = (kt < a.ktiles && ot0 + o < a.n_otiles) ? wpk[((int64_t)(ot0 + o) * a.ktiles + kt) * 64 + lane] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int o = 0; o < OT; ++o)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[o][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][o], xv[u][t], acc[o][t], 0, 0, 0);
    }
    if (!p_in) return;
    pw_finish<OT, VEC>(a, acc, ot0, k, b, p0);
}

// The same layer on the coarse maps (a few hundred workgroups at most): there a launch is a chain of K / 16 load-wait-multiply
// rounds per wave and most of the chip is idle, so the four waves of a workgroup split the K loop of ONE output tile (wave w
// takes rounds w, w + 4, ...) and the partial sums are added through LDS in wave order: a quarter of the dependent rounds
// (389 -> 273 channels over 135 positions: 33 -> ~12 us).  Summation order: channels of a wave's rounds ascending, then
// ((w0 + w1) + w2) + w3 -- fixed.
template <bool VEC>
__global__ __launch_bounds__(256) void pointwise_conv_ksplit_kernel(PwArgs a) {
    __shared__ float part[3][64][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = lane >> 4, j = lane & 15;
    const int ot = blockIdx.y;
    const int b = blockIdx.z;
    const int64_t P = a.P, p0 = (int64_t)blockIdx.x * 64 + 4 * j;
    const float *xb = a.x + (int64_t)b * a.x_bstride;
    const float *wpk = a.wpk + (int64_t)b * a.w_bstride;
    pw_f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = pw_f32x4{0.f, 0.f, 0.f, 0.f};
    const bool p_in = p0 < P;
    constexpr int U = 4;
    for (int kt0 = wave * U; kt0 < a.ktiles; kt0 += 4 * U) {
        float xv[U][4], av[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kt = kt0 + u, c = 4 * kt + k;
#pragma unroll
            for (int t = 0; t < 4; ++t) xv[u][t] = 0.f;
            if (c < a.Cin && p_in) {
                const float *row = xb + (int64_t)c * P + p0;
                if (VEC) {
                    const float4 v = *reinterpret_cast<const float4 *>(row);
                    xv[u][0] = v.x, xv[u][1] = v.y, xv[u][2] = v.z, xv[u][3] = v.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) xv[u][t] = p0 + t < P ? row[t] : 0.f;
                }
            }
            av[u] = kt < a.ktiles ? wpk[((int64_t)ot * a.ktiles + kt) * 64 + lane] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], xv[u][t], acc[t], 0, 0, 0);
    }
    // wave 0 finishes the tile: its affine pairs (and residual rows) are requested BEFORE the barrier, so that they arrive while the
    // other waves' sums do; activation as in pw_finish (one select per value, no branches)
    const int oc0 = 16 * ot + 4 * k;
    float sc[4] = {1.0f, 1.0f, 1.0f, 1.0f}, sh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float4 rr[4];
    const bool finisher = wave == 0 && p_in;
    const float *rb = a.res ? a.res + (int64_t)b * a.res_bstride + (int64_t)oc0 * P + p0 : nullptr;
    if (finisher) {
        if (a.scale) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[r] = a.scale[min(oc0 + r, a.Cout - 1)];
        }
        if (a.shift) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sh[r] = a.shift[min(oc0 + r, a.Cout - 1)];
        }
        if (VEC && rb) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (oc0 + r < a.Cout) rr[r] = *reinterpret_cast<const float4 *>(rb + (int64_t)r * P);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave - 1][lane][4 * t + r] = acc[t][r];
    }
    __syncthreads();
    if (!finisher) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] += part[w][lane][4 * t + r];
    const float alt_scale = a.act == 1 ? 0.0f : (a.act == 2 ? a.slope : 1.0f);
    const unsigned alt_keep = a.act == 1 ? 0u : ~0u;
    float *yb = a.y + ((int64_t)b * a.Cout + oc0) * P + p0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (oc0 + r >= a.Cout) continue;
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float u = sc[r] * acc[t][r] + sh[r];
            const float alt = __uint_as_float(__float_as_uint(u * alt_scale) & alt_keep);
            v[t] = u < 0.f ? alt : u;
        }
        const int64_t off = (int64_t)r * P;
        if (VEC) {
            if (rb) v[0] += rr[r].x, v[1] += rr[r].y, v[2] += rr[r].z, v[3] += rr[r].w;
            *reinterpret_cast<float4 *>(yb + off) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (p0 + t < P) yb[off + t] = rb ? v[t] + rb[off + t] : v[t];
        }
    }
}

template <int OT>
int launch_pw(const PwArgs &a, int B, bool vec, hipStream_t st) {
    dim3 grid((unsigned)((a.P + 63) / 64), (unsigned)((a.n_otiles + 4 * OT - 1) / (4 * OT)), (unsigned)B);
    if (vec) hipLaunchKernelGGL((pointwise_conv_kernel<OT, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pointwise_conv_kernel<OT, false>), grid, dim3(256), 0, st, a);
    return rpe_launch_status();
}

}  // namespace

RPE_API int rpe_pointwise_conv(const float *x, int64_t x_batch_stride, int B, int Cin, int64_t P, const float *packed_weight,
                               int64_t weight_batch_stride, int Cout, const float *scale, const float *shift, int act, float slope,
                               const float *residual, int64_t residual_batch_stride, float *y, rpe_stream_t stream) {
    if (!x || !packed_weight || !y || B < 0 || Cin < 1 || Cout < 1 || P < 0 || act < 0 || act > 2) return RPE_EINVAL;
    if (x_batch_stride < (int64_t)Cin * P || weight_batch_stride < 0 || (residual && residual_batch_stride < (int64_t)Cout * P)) return RPE_EINVAL;
    if (B == 0 || P == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    PwArgs a{x, packed_weight, scale, shift, residual, y, Cin, Cout, (Cin + 3) / 4, (Cout + 15) / 16, act, P, x_batch_stride, weight_batch_stride, slope, residual_batch_stride};
    const bool vec = P % 4 == 0 && x_batch_stride % 4 == 0 && residual_batch_stride % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
    hipStream_t st = (hipStream_t)stream;
    // output tiles per wave: enough workgroups to cover the chip on the small maps, fewer re-reads of x on the wider layers
    const int64_t wgs1 = ((P + 63) / 64) * ((a.n_otiles + 3) / 4) * B;
#ifndef RPE_PW_KSPLIT_MAX_WGS
#define RPE_PW_KSPLIT_MAX_WGS 1024
#endif
    const int64_t wgs_ks = ((P + 63) / 64) * a.n_otiles * B;
    if (wgs_ks <= RPE_PW_KSPLIT_MAX_WGS && a.ktiles >= 16 && a.n_otiles <= 65535) {
        dim3 grid((unsigned)((P + 63) / 64), (unsigned)a.n_otiles, (unsigned)B);
        if (vec) hipLaunchKernelGGL((pointwise_conv_ksplit_kernel<true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((pointwise_conv_ksplit_kernel<false>), grid, dim3(256), 0, st, a);
        return rpe_launch_status();
    }
#ifndef RPE_PW_OT4_MIN_TILES
#define RPE_PW_OT4_MIN_TILES 16
#endif
#ifndef RPE_PW_OT2_MIN_TILES
#define RPE_PW_OT2_MIN_TILES 8
#endif
#ifndef RPE_PW_OT2_MIN_WGS
#define RPE_PW_OT2_MIN_WGS 2048
#endif
#ifndef RPE_PW_OT4_MIN_WGS
#define RPE_PW_OT4_MIN_WGS 8192  // (72 x 120 maps, 340 / 510 output channels: two tiles a wave 48 / 49 us, four 53 / 56)
#endif
    if (a.n_otiles >= RPE_PW_OT4_MIN_TILES && wgs1 >= RPE_PW_OT4_MIN_WGS) return launch_pw<4>(a, B, vec, st);
    if (a.n_otiles >= RPE_PW_OT2_MIN_TILES && wgs1 >= RPE_PW_OT2_MIN_WGS) return launch_pw<2>(a, B, vec, st);
    // 5 .. 7 output tiles from many input channels: three waves of two tiles read x once instead of twice (tools/pw_census.py,
    // 144 x 240: 255 -> 96 108 -> 103 us, 215 -> 81 94 -> 86, 113 -> 81 53 -> 51; below ~100 input channels the same rule loses:
    // 96 -> 96 46 -> 51, 67 -> 96 36 -> 39)
    if (a.n_otiles >= 5 && a.ktiles >= 28 && wgs1 >= RPE_PW_OT2_MIN_WGS) return launch_pw<2>(a, B, vec, st);
    return launch_pw<1>(a, B, vec, st);
}
