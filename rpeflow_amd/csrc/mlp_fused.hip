// One- or two-layer point-wise MLP (1x1 Conv1d + bias / eval-BatchNorm + activation per layer: Conv1dNormRelu / MLP1d,
// models/utils.py:7-98) as ONE kernel for gfx950, for the small layers between the 3-D hot-path operators
// (pwc3d_core.py:36-41, 125; RPEFlow_core.py:378-391): each was a library GEMM launch plus an epilogue launch.
//
// Transposed formulation, so that a chain of layers stays in the D-register layout: 16 points a workgroup, and
//     h^T[c1][n] = act1(W1[c1][:] . x[:, n])          v_mfma_f32_16x16x4_f32: A = W1 fragment, B = x[c][n] read channel-first
//     y^T[c2][n] = act2(W2[c2][:] . h[:, n])          A = W2 fragment, B = the D registers of layer 1 AS THEY ARE:
// lane (kk, n) holds h rows 16t + 4kk + r in register r of tile t, which is exactly the k-slot kk / step r operand of a
// 16x16x4 step when the W2 fragments are packed with that k order (done once per module on the host).
// Output channel-first [B,C,N], or channel-last rows [xyz | y | 0] -- the PointConv kernel's gather source -- which
// replaces the packing pass behind the pyramid's MLPs.  fp32; weights come from L2 (every wave streams all of them).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MlpArgs {
    const float *x;
    int64_t x_sb, x_sc, x_sn;
    int C0, N;
    const f32x4 *w1p;  // [ceil(C0/16)][T1][64]: lane (row o = l & 15, kk = l >> 4), element s = W1[16 t + o][16 g + 4 kk + s]
    const f32x4 *ss1;  // [4 T1][2]... stored as [T1*4] float4 scale then [T1*4] float4 shift, index 4 t + kk: channels 16 t + 4 kk + r
    int act1;
    const f32x4 *w2p;  // [T1][T2][64]: element s = W2[16 t2 + o][16 t1 + 4 kk + s]
    const f32x4 *ss2;
    int act2;
    float slope;
    float *out;
    int out_mode, out_stride, Cout;  // 0: [B,Cout,N]; 1: rows [B,N,out_stride] = [xyz | y | zeros]
    const float *xyz;
    int64_t z_sb, z_sd, z_sn;
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ f32x4 epilogue(f32x4 v, f32x4 scale, f32x4 shift, int act, float slope) {
    f32x4 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = v[r] * scale[r] + shift[r];
        if (act == 1) t = rpe_relu(t);
        if (act == 2) t = t >= 0.f ? t : t * slope;
        y[r] = t;
    }
    return y;
}

// Narrow layers (at most 32 channels a layer): a wave owns 16 points and keeps the chain of layers in its registers.
template <int T1, int T2>
__global__ __launch_bounds__(256) void mlp_wave_kernel(MlpArgs a) {
    const int lane = rpe_lane(), kk = lane >> 4, n16 = lane & 15;
    const int wave = rpe_uniform((int)(threadIdx.x >> 6));
    const int b = blockIdx.y, n0 = (blockIdx.x * 4 + wave) * 16;
    if (n0 >= a.N) return;
    const int n = min(n0 + n16, a.N - 1);
    const float *xp = a.x + (int64_t)b * a.x_sb + (int64_t)n * a.x_sn;

    f32x4 h[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t) h[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int groups = (a.C0 + 15) / 16;
    for (int g = 0; g < groups; ++g) {
        float xv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = 16 * g + 4 * kk + s;
            xv[s] = c < a.C0 ? xp[(int64_t)c * a.x_sc] : 0.f;
        }
        const f32x4 *wf = a.w1p + (int64_t)g * T1 * 64 + lane;
#pragma unroll
        for (int t = 0; t < T1; ++t) {
            const f32x4 w = wf[t * 64];
#pragma unroll
            for (int s = 0; s < 4; ++s) h[t] = mfma16(w[s], xv[s], h[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < T1; ++t) h[t] = epilogue(h[t], a.ss1[4 * t + kk], a.ss1[4 * T1 + 4 * t + kk], a.act1, a.slope);

    constexpr int TO = T2 > 0 ? T2 : T1;
    f32x4 y[TO];
    if constexpr (T2 > 0) {
#pragma unroll
        for (int t2 = 0; t2 < T2; ++t2) y[t2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t1 = 0; t1 < T1; ++t1) {
            const f32x4 *wf = a.w2p + (int64_t)t1 * T2 * 64 + lane;
#pragma unroll
            for (int t2 = 0; t2 < T2; ++t2) {
                const f32x4 w = wf[t2 * 64];
#pragma unroll
                for (int s = 0; s < 4; ++s) y[t2] = mfma16(w[s], h[t1][s], y[t2]);
            }
        }
#pragma unroll
        for (int t2 = 0; t2 < T2; ++t2) y[t2] = epilogue(y[t2], a.ss2[4 * t2 + kk], a.ss2[4 * T2 + 4 * t2 + kk], a.act2, a.slope);
    } else {
#pragma unroll
        for (int t = 0; t < T1; ++t) y[t] = h[t];
    }

    // lane (kk, n16), tile t, register r: output channel 16 t + 4 kk + r of point n0 + n16
    if (n0 + n16 >= a.N) return;
    if (a.out_mode == 0) {
        float *o = a.out + (int64_t)b * a.Cout * a.N + (n0 + n16);
#pragma unroll
        for (int t = 0; t < TO; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * t + 4 * kk + r;
                if (c < a.Cout) o[(int64_t)c * a.N] = y[t][r];
            }
    } else {
        float *row = a.out + ((int64_t)b * a.N + n0 + n16) * a.out_stride;
#pragma unroll
        for (int t = 0; t < TO; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * t + 4 * kk + r;
                if (3 + c < a.out_stride) row[3 + c] = c < a.Cout ? y[t][r] : 0.f;
            }
        if (kk == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) row[d] = a.xyz[(int64_t)b * a.z_sb + d * a.z_sd + (int64_t)(n0 + n16) * a.z_sn];
            for (int z = 3 + 16 * TO; z < a.out_stride; ++z) row[z] = 0.f;
        }
    }
}

// A workgroup of four waves owns 16 points; wave w computes the output tiles t = w, w + 4, ... of each layer, so a layer's
// weight fragments are read once per workgroup and a wave's MFMA chain is a quarter of the layer.  (One wave per 16 points
// took ~18 us whatever the point count: 384 dependent-ish MFMAs and 96 KB of fragments streamed by every wave.)  Layer 1's
// tiles meet in LDS -- as the f32x4 D registers they are, which is the B operand layout of layer 2 -- behind one barrier.
template <int T1, int T2>  // output tiles of layer 1 and layer 2 (T2 = 0: one layer)
__global__ __launch_bounds__(256) void mlp_fused_kernel(MlpArgs a) {
    constexpr int P1 = (T1 + 3) / 4, P2 = (T2 + 3) / 4;  // tiles per wave
    __shared__ f32x4 hx[T2 > 0 ? T1 : 1][64];
    const int lane = rpe_lane(), kk = lane >> 4, n16 = lane & 15;
    const int wave = rpe_uniform((int)(threadIdx.x >> 6));
    const int b = blockIdx.y, n0 = blockIdx.x * 16;
    const int n = min(n0 + n16, a.N - 1);
    const float *xp = a.x + (int64_t)b * a.x_sb + (int64_t)n * a.x_sn;

    // the input values and the weights of a K-group are requested one group ahead of the MFMAs that use them
    f32x4 h[P1];
#pragma unroll
    for (int p = 0; p < P1; ++p) h[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int groups = (a.C0 + 15) / 16;
    auto load_x = [&](int g, float (&xv)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = 16 * g + 4 * kk + s;
            xv[s] = (g < groups && c < a.C0) ? xp[(int64_t)c * a.x_sc] : 0.f;
        }
    };
    auto load_w1 = [&](int g, f32x4 (&w)[P1]) {
        const f32x4 *wf = a.w1p + (int64_t)min(g, groups - 1) * T1 * 64 + lane;
#pragma unroll
        for (int p = 0; p < P1; ++p) w[p] = wf[min(wave + 4 * p, T1 - 1) * 64];
    };
    float xa[4], xb[4];
    f32x4 wa[P1], wb[P1];
    load_x(0, xa);
    load_w1(0, wa);
    for (int g = 0; g < groups; g += 2) {
        load_x(g + 1, xb);
        load_w1(g + 1, wb);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < P1; ++p) h[p] = mfma16(wa[p][s], xa[s], h[p]);
        if (g + 1 >= groups) break;
        load_x(g + 2, xa);
        load_w1(g + 2, wa);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < P1; ++p) h[p] = mfma16(wb[p][s], xb[s], h[p]);
    }
    constexpr int PO = T2 > 0 ? P2 : P1, TO = T2 > 0 ? T2 : T1;
    f32x4 w2a[T2 > 0 ? P2 : 1], w2b[T2 > 0 ? P2 : 1];
    auto load_w2 = [&](int t1, f32x4 (&w)[T2 > 0 ? P2 : 1]) {
        const f32x4 *wf = a.w2p + (int64_t)min(t1, T1 - 1) * T2 * 64 + lane;
#pragma unroll
        for (int p = 0; p < P2; ++p) w[p] = wf[min(wave + 4 * p, T2 - 1) * 64];
    };
    if constexpr (T2 > 0) load_w2(0, w2a);
#pragma unroll
    for (int p = 0; p < P1; ++p) {
        const int t = wave + 4 * p;
        if (t < T1) h[p] = epilogue(h[p], a.ss1[4 * t + kk], a.ss1[4 * T1 + 4 * t + kk], a.act1, a.slope);
    }

    f32x4 y[PO];
    if constexpr (T2 > 0) {
#pragma unroll
        for (int p = 0; p < P1; ++p)
            if (wave + 4 * p < T1) hx[wave + 4 * p][lane] = h[p];
        __syncthreads();
#pragma unroll
        for (int p = 0; p < P2; ++p) y[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t1 = 0; t1 < T1; ++t1) {
            f32x4 (&cur)[P2] = (t1 & 1) ? w2b : w2a;
            f32x4 (&nxt)[P2] = (t1 & 1) ? w2a : w2b;
            if (t1 + 1 < T1) load_w2(t1 + 1, nxt);
            const f32x4 hv = hx[t1][lane];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int p = 0; p < P2; ++p) y[p] = mfma16(cur[p][s], hv[s], y[p]);
        }
#pragma unroll
        for (int p = 0; p < P2; ++p) {
            const int t2 = wave + 4 * p;
            if (t2 < T2) y[p] = epilogue(y[p], a.ss2[4 * t2 + kk], a.ss2[4 * T2 + 4 * t2 + kk], a.act2, a.slope);
        }
    } else {
#pragma unroll
        for (int p = 0; p < P1; ++p) y[p] = h[p];
    }

    // lane (kk, n16), tile t, register r: output channel 16 t + 4 kk + r of point n0 + n16
    if (n0 + n16 >= a.N) return;
    if (a.out_mode == 0) {
        float *o = a.out + (int64_t)b * a.Cout * a.N + (n0 + n16);
#pragma unroll
        for (int p = 0; p < PO; ++p) {
            const int t = wave + 4 * p;
            if (t >= TO) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * t + 4 * kk + r;
                if (c < a.Cout) o[(int64_t)c * a.N] = y[p][r];
            }
        }
    } else {
        float *row = a.out + ((int64_t)b * a.N + n0 + n16) * a.out_stride;
#pragma unroll
        for (int p = 0; p < PO; ++p) {
            const int t = wave + 4 * p;
            if (t >= TO) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * t + 4 * kk + r;
                if (3 + c < a.out_stride) row[3 + c] = c < a.Cout ? y[p][r] : 0.f;
            }
        }
        if (wave == 0 && kk == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) row[d] = a.xyz[(int64_t)b * a.z_sb + d * a.z_sd + (int64_t)(n0 + n16) * a.z_sn];
            for (int z = 3 + 16 * TO; z < a.out_stride; ++z) row[z] = 0.f;
        }
    }
}

template <int T1, int T2>
int launch(const MlpArgs &a, int B, hipStream_t st) {
    if constexpr (T1 <= 2 && T2 <= 2)  // too few tiles to split over four waves
        hipLaunchKernelGGL((mlp_wave_kernel<T1, T2>), dim3((a.N + 63) / 64, B), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((mlp_fused_kernel<T1, T2>), dim3((a.N + 15) / 16, B), dim3(256), 0, st, a);
    return rpe_launch_status();
}

}  // namespace

RPE_API int rpe_mlp1d_fused(const float *x, int64_t x_sb, int64_t x_sc, int64_t x_sn, int B, int C0, int N, const float *w1_packed,
                            const float *scale_shift1, int act1, int T1, const float *w2_packed, const float *scale_shift2, int act2,
                            int T2, float slope, int Cout, int out_mode, int out_stride, const float *xyz, int64_t z_sb, int64_t z_sd,
                            int64_t z_sn, float *out, rpe_stream_t stream) {
    if (!x || !w1_packed || !scale_shift1 || !out || B < 0 || C0 < 1 || N < 0 || T1 < 1 || T2 < 0 || Cout < 1) return RPE_EINVAL;
    if (T2 > 0 && (!w2_packed || !scale_shift2)) return RPE_EINVAL;
    if (act1 < 0 || act1 > 2 || act2 < 0 || act2 > 2 || Cout > 16 * (T2 > 0 ? T2 : T1)) return RPE_EINVAL;
    if (out_mode != 0 && (out_mode != 1 || !xyz || out_stride < 3 + Cout)) return RPE_EINVAL;
    if (B == 0 || N == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    MlpArgs a{x, x_sb, x_sc, x_sn, C0, N, (const f32x4 *)w1_packed, (const f32x4 *)scale_shift1, act1, (const f32x4 *)w2_packed,
              (const f32x4 *)scale_shift2, act2, slope, out, out_mode, out_stride, Cout, xyz, z_sb, z_sd, z_sn};
    hipStream_t st = (hipStream_t)stream;
#define RPE_MLP_CASE(A, Bt) \
    if (T1 == A && T2 == Bt) return launch<A, Bt>(a, B, st);
    // the layer shapes of the 3-D branch (channels / 16): pyramid MLPs C_i -> C_i -> C_{i+1}, the estimator's 128 -> 128 -> 64,
    // single layers up to 192 outputs
    RPE_MLP_CASE(1, 1) RPE_MLP_CASE(1, 2) RPE_MLP_CASE(2, 4) RPE_MLP_CASE(4, 6) RPE_MLP_CASE(6, 8) RPE_MLP_CASE(8, 12) RPE_MLP_CASE(8, 4)
    RPE_MLP_CASE(1, 0) RPE_MLP_CASE(2, 0) RPE_MLP_CASE(4, 0) RPE_MLP_CASE(6, 0) RPE_MLP_CASE(8, 0) RPE_MLP_CASE(12, 0)
#undef RPE_MLP_CASE
    return RPE_EUNSUPPORTED;
}
