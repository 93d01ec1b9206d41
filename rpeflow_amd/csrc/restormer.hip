// Memory-bound pieces of the Restormer cross-attention blocks (models/restormer_arch.py), SURVEY.md
// section 8(f) rank 1 -- the first widening beyond the hot path.  On the GPU the reference runs each of
// these as a chain of PyTorch kernels; MIOpen has no tuned depth-wise 3x3 fp32 solver for these shapes and
// falls back to naive_conv (65 us a call, 80 calls a forward).
//
//   dwconv3_kernel      depth-wise 3x3 (2-D) / 3-tap (1-D) convolution, stride 1, zero padding 1, optional bias
//                       (qkv_dwconv, restormer_arch.py:175-176, 256-257; dwconv :96-97, 234-235).  The input may be
//                       given as up to three channel segments, which fuses torch.cat((x, y, y)) (:182, 263);
//                       gate = 1 fuses the GDFN gate gelu(x1) * x2 over the two halves of the channels (:104-105, 244-245).
//   channel_norm_kernel LayerNorm over the channel axis of [B,C,P] (WithBias / BiasFree, :31-63), C <= 256: x read once
//                       into registers, exact two-pass mean / biased variance, normalise + affine.
#include <math.h>

#include "common.h"

namespace {

struct Segments {
    const float *ptr[3];
    int ch[3];  // channels per segment; sum = C
};

__device__ __forceinline__ const float *plane(const Segments &s, int b, int c, int64_t HW) {
    if (c < s.ch[0]) return s.ptr[0] + ((int64_t)b * s.ch[0] + c) * HW;
    c -= s.ch[0];
    if (c < s.ch[1]) return s.ptr[1] + ((int64_t)b * s.ch[1] + c) * HW;
    c -= s.ch[1];
    return s.ptr[2] + ((int64_t)b * s.ch[2] + c) * HW;
}

template <int KH>  // 3: 2-D 3x3, 1: 1-D 3-tap (H == 1)
__device__ __forceinline__ float dw_at(const float *in, const float *w, int y, int x, int H, int W) {
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < KH; ++ky) {
        const int yy = y + ky - KH / 2;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1;
            if (xx >= 0 && xx < W) s = __fmaf_rn(w[ky * 3 + kx], in[(int64_t)yy * W + xx], s);
        }
    }
    return s;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int KH, bool GATE>
__global__ __launch_bounds__(256) void dwconv3_kernel(Segments seg, const float *__restrict__ weight, const float *__restrict__ bias,
                                                      int C, int H, int W, float *__restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    const int64_t HW = (int64_t)H * W;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const int Cout = GATE ? C / 2 : C;
    float v = dw_at<KH>(plane(seg, b, c, HW), weight + (int64_t)c * KH * 3, y, x, H, W);
    if (bias) v += bias[c];
    if (GATE) {
        const int c2 = c + Cout;
        float g = dw_at<KH>(plane(seg, b, c2, HW), weight + (int64_t)c2 * KH * 3, y, x, H, W);
        if (bias) g += bias[c2];
        v = gelu_erf(v) * g;
    }
    out[((int64_t)b * Cout + c) * HW + p] = v;
}

// Strip form for W % 4 == 0: one thread produces a 4-wide x R-high patch of one channel plane.  Every input row is
// fetched once as a float4 plus its two halo scalars and feeds up to KH output rows from registers, so a patch costs
// (R + KH - 1) * 3 load instructions for 4R outputs instead of 9 per output.
template <int KH, int R>
__device__ __forceinline__ void dw_strip(const float *__restrict__ in, const float *__restrict__ w, int y0, int x, int H, int W,
                                         float (&acc)[R][4]) {
    float wv[KH * 3];
#pragma unroll
    for (int i = 0; i < KH * 3; ++i) wv[i] = w[i];
#pragma unroll
    for (int o = 0; o < R; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[o][i] = 0.f;
#pragma unroll
    for (int r = -(KH / 2); r < R + KH / 2; ++r) {
        const int yy = y0 + r;
        if (yy < 0 || yy >= H) continue;
        const float *row = in + (int64_t)yy * W + x;
        const float4 m = *reinterpret_cast<const float4 *>(row);
        const float v[6] = {x > 0 ? row[-1] : 0.f, m.x, m.y, m.z, m.w, x + 4 < W ? row[4] : 0.f};
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
            const int o = r - ky + KH / 2;  // output row fed by this input row through tap ky
            if (o < 0 || o >= R) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[o][i] = __fmaf_rn(wv[ky * 3 + kx], v[i + kx], acc[o][i]);
        }
    }
}

template <int KH, bool GATE>
__global__ __launch_bounds__(256) void dwconv3_strip_kernel(Segments seg, const float *__restrict__ weight,
                                                            const float *__restrict__ bias, int C, int H, int W,
                                                            float *__restrict__ out) {
    constexpr int R = KH == 3 ? 4 : 1;
    const int W4 = W >> 2, HR = (H + R - 1) / R;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W4 * HR) return;
    const int yb = t / W4, x = (t - yb * W4) * 4, y0 = yb * R;
    const int c = blockIdx.y, b = blockIdx.z;
    const int64_t HW = (int64_t)H * W;
    const int Cout = GATE ? C / 2 : C;
    float acc[R][4];
    dw_strip<KH, R>(plane(seg, b, c, HW), weight + (int64_t)c * KH * 3, y0, x, H, W, acc);
    const float b0 = bias ? bias[c] : 0.f;
    if (GATE) {
        const int c2 = c + Cout;
        float gat[R][4];
        dw_strip<KH, R>(plane(seg, b, c2, HW), weight + (int64_t)c2 * KH * 3, y0, x, H, W, gat);
        const float b1 = bias ? bias[c2] : 0.f;
#pragma unroll
        for (int o = 0; o < R; ++o)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[o][i] = gelu_erf(acc[o][i] + b0) * (gat[o][i] + b1);
    } else {
#pragma unroll
        for (int o = 0; o < R; ++o)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[o][i] += b0;
    }
    float *op = out + ((int64_t)b * Cout + c) * HW + (int64_t)y0 * W + x;
#pragma unroll
    for (int o = 0; o < R; ++o)
        if (y0 + o < H) *reinterpret_cast<float4 *>(op + (int64_t)o * W) = make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]);
}

// ---- the tail of the gated feed-forward in ONE launch: depth-wise 3x3 (or 3-tap) + bias + gelu gate + project_out (1x1) + bias
// + residual (restormer_arch.py:104-106 / 244-246; FeedForward.forward after project_in).  The gated tensor g [B, hidden, P] is
// the B operand of project_out's matrix product; here it goes from the lanes that compute it to the lanes that multiply it
// through LDS instead of HBM (2 x 119 MB of 850 MB per level-1 block, and a launch).  A workgroup owns 64 positions; a round is
// 16 hidden channels: wave w computes the gate of channel group 4 r + w in the 1x1 kernel's operand layout (lane (k, j): channel
// 4 kt + k, positions p0 .. p0 + 3 of one map row; dw_strip and gelu_erf exactly as dwconv3_strip_kernel<KH, true> runs them)
// and leaves it in LDS for everybody; then every wave multiplies the round's four groups into ITS output tiles (w and w + 4: up
// to 128 output channels) in the channel order of pointwise_conv_kernel.  The gate of round r + 1 is computed before the
// products of round r are issued (double-buffered: one barrier a round).  (A first form -- one wave per 64 positions computing
// every gate itself and holding all output tiles -- was 282 us against the two launches' 236: 204 VGPRs, two waves a SIMD, a
// 64-round serial loop.)
typedef float gd_f32x4 __attribute__((ext_vector_type(4)));

template <int KH>
__global__ __launch_bounds__(256) void gdfn_tail_kernel(const float *__restrict__ t, int hidden, int H, int W, const float *__restrict__ dw_w,
                                                        const float *__restrict__ dw_b, const float *__restrict__ wpk, int ktiles,
                                                        int n_otiles, int Cout, const float *__restrict__ shift, const float *res, float *y) {
    __shared__ float4 gbuf[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = lane >> 4, j = lane & 15;
    const int b = blockIdx.y;
    const int64_t P = (int64_t)H * W;
    const int64_t p0 = (int64_t)blockIdx.x * 64 + 4 * j;
    const bool p_in = p0 < P;  // (W % 4 == 0: the four positions share a row and are in range together)
    const int py = p_in ? (int)(p0 / W) : 0, px = p_in ? (int)(p0 - (int64_t)py * W) : 0;
    const float *tb = t + (int64_t)b * 2 * hidden * P;
    gd_f32x4 acc[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[o][q] = gd_f32x4{0.f, 0.f, 0.f, 0.f};
    auto gate = [&](int kt) {  // this wave's channel group of a round
        float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = 4 * kt + k;
        if (kt < ktiles && c < hidden && p_in) {
            float a[1][4], g[1][4];
            dw_strip<KH, 1>(tb + (int64_t)c * P, dw_w + (int64_t)c * KH * 3, py, px, H, W, a);
            dw_strip<KH, 1>(tb + (int64_t)(hidden + c) * P, dw_w + (int64_t)(hidden + c) * KH * 3, py, px, H, W, g);
            const float b0 = dw_b ? dw_b[c] : 0.f, b1 = dw_b ? dw_b[hidden + c] : 0.f;
            gv.x = gelu_erf(a[0][0] + b0) * (g[0][0] + b1), gv.y = gelu_erf(a[0][1] + b0) * (g[0][1] + b1);
            gv.z = gelu_erf(a[0][2] + b0) * (g[0][2] + b1), gv.w = gelu_erf(a[0][3] + b0) * (g[0][3] + b1);
        }
        return gv;
    };
    const int rounds = (ktiles + 3) / 4;
    const int o0 = wave, o1 = wave + 4;
    gbuf[0][wave][lane] = gate(wave);
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        float av[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kt = 4 * r + u;
            av[u][0] = (kt < ktiles && o0 < n_otiles) ? wpk[((int64_t)o0 * ktiles + kt) * 64 + lane] : 0.f;
            av[u][1] = (kt < ktiles && o1 < n_otiles) ? wpk[((int64_t)o1 * ktiles + kt) * 64 + lane] : 0.f;
        }
        const float4 nxt = r + 1 < rounds ? gate(4 * (r + 1) + wave) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 g = gbuf[r & 1][u][lane];
            const float gq[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][0], gq[q], acc[0][q], 0, 0, 0);
                acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][1], gq[q], acc[1][q], 0, 0, 0);
            }
        }
        gbuf[(r + 1) & 1][wave][lane] = nxt;
        __syncthreads();
    }
    if (!p_in) return;
    // D layout: lane (g = lane >> 4, j), register r: output 16 o + 4 g + r, position p0 + q of tile q
#pragma unroll
    for (int oo = 0; oo < 2; ++oo) {
        const int o = oo == 0 ? o0 : o1;
        if (o >= n_otiles) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = 16 * o + 4 * k + r;
            if (oc >= Cout) continue;
            const float sh = shift ? shift[oc] : 0.0f;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = acc[oo][q][r] + sh;
            const int64_t off = ((int64_t)b * Cout + oc) * P + p0;
            if (res) {
                const float4 rr = *reinterpret_cast<const float4 *>(res + off);
                v[0] += rr.x, v[1] += rr.y, v[2] += rr.z, v[3] += rr.w;
            }
            *reinterpret_cast<float4 *>(y + off) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// 64 positions x 4 channel groups per workgroup: thread (pl, cg) keeps channels cg, cg+4, ... of its position in
// registers (x is read once), the two reductions over the four groups (mean, then sum of squared deviations:
// the exact two-pass variance) go through LDS.
// Up to two independent tensors of one shape per launch (blockIdx.z): the cross blocks normalise x and y side by side.
struct NormJobs {
    const float *x[2], *weight[2], *bias[2];
    float *out[2];
};

template <int CPT>  // channels per thread: C <= 4 * CPT
__global__ __launch_bounds__(256) void channel_norm_kernel(NormJobs jobs, int C, int64_t P, float eps) {
    const float *__restrict__ x = jobs.x[blockIdx.z];
    const float *__restrict__ weight = jobs.weight[blockIdx.z];
    const float *__restrict__ bias = jobs.bias[blockIdx.z];
    float *__restrict__ out = jobs.out[blockIdx.z];
    __shared__ float part[2][4][64];
    const int pl = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 64 + pl;
    const int b = blockIdx.y;
    const bool live = p < P;
    const float *xb = x + (int64_t)b * C * P + (live ? p : 0);
    float v[CPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = cg + 4 * i;
        v[i] = (live && c < C) ? xb[(int64_t)c * P] : 0.f;
        s += v[i];
    }
    part[0][cg][pl] = s;
    __syncthreads();
    const float mean = (part[0][0][pl] + part[0][1][pl] + part[0][2][pl] + part[0][3][pl]) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = cg + 4 * i;
        const float d = v[i] - mean;
        q += (c < C) ? d * d : 0.f;
    }
    part[1][cg][pl] = q;
    __syncthreads();
    const float var = (part[1][0][pl] + part[1][1][pl] + part[1][2][pl] + part[1][3][pl]) / (float)C;
    const float inv = 1.0f / sqrtf(var + eps);
    if (!live) return;
    float *ob = out + (int64_t)b * C * P + p;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = cg + 4 * i;
        if (c < C) ob[(int64_t)c * P] = bias ? (v[i] - mean) * inv * weight[c] + bias[c] : v[i] * inv * weight[c];
    }
}

// y[b][c][p] = act(scale[c] * y + shift[c]), in place: the bias add, eval-mode BatchNorm and activation that
// follow every Conv{1,2}dNormRelu convolution (models/utils.py:7-62) as one pass instead of three kernels.
// act: 0 none, 1 relu, 2 leaky_relu(slope)
__global__ __launch_bounds__(256) void affine_act_kernel(float *__restrict__ y, const float *__restrict__ scale,
                                                         const float *__restrict__ shift, int C, int64_t P, int act, float slope) {
    const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int c = blockIdx.y, b = blockIdx.z;
    if (p >= P) return;
    const float a = scale ? scale[c] : 1.0f, s = shift ? shift[c] : 0.0f;
    float *row = y + ((int64_t)b * C + c) * P + p;
    auto f = [&](float v) {
        v = a * v + s;
        return act == 1 ? rpe_relu(v) : (act == 2 ? (v >= 0.f ? v : v * slope) : v);
    };
    if (p + 4 <= P && ((reinterpret_cast<uintptr_t>(row) & 15) == 0)) {
        float4 v = *reinterpret_cast<float4 *>(row);
        v.x = f(v.x); v.y = f(v.y); v.z = f(v.z); v.w = f(v.w);
        *reinterpret_cast<float4 *>(row) = v;
    } else {
        for (int i = 0; i < 4 && p + i < P; ++i) row[i] = f(row[i]);
    }
}

// y[b][c][p] = act(scale[c] * y + shift[c] + zscale[c] * z[b][c][p]), in place on y: the tail of a residual block --
// act(BN(conv1(..)) + BN(down0(x))), pwc2d_core.py:6-25 -- as one pass over the two raw convolution outputs instead of two
// epilogue passes, an add and the activation (shift = both branches' shifts, summed by the caller).
__global__ __launch_bounds__(256) void affine_add_act_kernel(float *__restrict__ y, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, const float *__restrict__ z,
                                                             const float *__restrict__ zscale, int C, int64_t P, int act, float slope) {
    const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int c = blockIdx.y, b = blockIdx.z;
    if (p >= P) return;
    const float a = scale ? scale[c] : 1.0f, s = shift ? shift[c] : 0.0f, za = zscale ? zscale[c] : 1.0f;
    float *row = y + ((int64_t)b * C + c) * P + p;
    const float *zrow = z + ((int64_t)b * C + c) * P + p;
    auto f = [&](float v, float w) {
        v = (a * v + s) + za * w;
        return act == 1 ? rpe_relu(v) : (act == 2 ? (v >= 0.f ? v : v * slope) : v);
    };
    if (p + 4 <= P && (((reinterpret_cast<uintptr_t>(row) | reinterpret_cast<uintptr_t>(zrow)) & 15) == 0)) {
        float4 v = *reinterpret_cast<float4 *>(row);
        const float4 w = *reinterpret_cast<const float4 *>(zrow);
        v.x = f(v.x, w.x); v.y = f(v.y, w.y); v.z = f(v.z, w.z); v.w = f(v.w, w.w);
        *reinterpret_cast<float4 *>(row) = v;
    } else {
        for (int i = 0; i < 4 && p + i < P; ++i) row[i] = f(row[i], zrow[i]);
    }
}

// The same tail with the shortcut branch computed here: y = act(scale * y + shift + zscale * (W0 . x[:, :, ::S, ::S])).
// The reference's down-sampling shortcut is a 1x1 convolution of stride 2 without padding (pwc2d_core.py:9: kernel_size 1,
// stride 2): per output pixel a Cin-long dot product with a row of W0 -- 0.05 to 0.3 GFLOP per pyramid level, nothing for
// the vector units, but as its own layer it was a strided copy + a GEMM launch (or a MIOpen convolution at the finest level)
// in front of this pass, on the main chain of both 2-D pyramids.  A thread owns one output pixel and kTailOc output channels;
// the weights of the block's channel group sit in LDS (read at one address by the whole wave: broadcast).  Summation over
// the input channels in ascending order, one chain per output: fixed.
constexpr int kTailOc = 8;
constexpr int kTailMaxCin = 256;
__global__ __launch_bounds__(256) void residual_tail_kernel(float *__restrict__ y, const float *__restrict__ scale, const float *__restrict__ shift,
                                                            const float *__restrict__ x, const float *__restrict__ w0, const float *__restrict__ zscale,
                                                            int Cin, int Cout, int H, int W, int Ho, int Wo, int stride, int act, float slope) {
    __shared__ float wl[kTailOc][kTailMaxCin];
    const int oc0 = blockIdx.y * kTailOc, b = blockIdx.z;
    for (int i = threadIdx.x; i < kTailOc * Cin; i += blockDim.x) {
        const int o = i / Cin, c = i - o * Cin;
        wl[o][c] = oc0 + o < Cout ? w0[(int64_t)(oc0 + o) * Cin + c] : 0.f;
    }
    __syncthreads();
    const int P = Ho * Wo;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int oy = p / Wo, ox = p - oy * Wo;
    const float *xp = x + (int64_t)b * Cin * H * W + (int64_t)(oy * stride) * W + ox * stride;
    float acc[kTailOc];
#pragma unroll
    for (int o = 0; o < kTailOc; ++o) acc[o] = 0.f;
    const int64_t plane = (int64_t)H * W;
    int c = 0;
    for (; c + 4 <= Cin; c += 4) {  // four loads in flight
        const float v0 = xp[(int64_t)c * plane], v1 = xp[(int64_t)(c + 1) * plane], v2 = xp[(int64_t)(c + 2) * plane], v3 = xp[(int64_t)(c + 3) * plane];
#pragma unroll
        for (int o = 0; o < kTailOc; ++o) acc[o] = fmaf(wl[o][c + 3], v3, fmaf(wl[o][c + 2], v2, fmaf(wl[o][c + 1], v1, fmaf(wl[o][c], v0, acc[o]))));
    }
    for (; c < Cin; ++c) {
        const float v = xp[(int64_t)c * plane];
#pragma unroll
        for (int o = 0; o < kTailOc; ++o) acc[o] = fmaf(wl[o][c], v, acc[o]);
    }
#pragma unroll
    for (int o = 0; o < kTailOc; ++o) {
        const int oc = oc0 + o;
        if (oc >= Cout) break;
        const float a = scale ? scale[oc] : 1.0f, s = shift ? shift[oc] : 0.0f, za = zscale ? zscale[oc] : 1.0f;
        float *row = y + ((int64_t)b * Cout + oc) * P + p;
        float v = (a * *row + s) + za * acc[o];
        *row = act == 1 ? rpe_relu(v) : (act == 2 ? (v >= 0.f ? v : v * slope) : v);
    }
}

// RAFT-style convex up-sampling (models/utils.py:201-214, the last step of RPEFlow_core.forward :424): every fine pixel
// (h*s+i, w*s+j) is a softmax-weighted combination (9 weights from mask channels k*s*s + i*s + j) of the 3x3
// coarse neighbourhood of s*flow.  One thread per coarse pixel walks its s*s fine pixels: the mask planes are read
// fully coalesced, once; the reference's softmax / unfold / product / sum / permute chain and its [B,2,9,s,s,H,W]
// intermediates are gone.
template <int S>
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float *__restrict__ flow, const float *__restrict__ mask, int H, int W,
                                                              float *__restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    const int HW = H * W;
    if (p >= HW) return;
    const int h = p / W, w = p - h * W;
    float f[2][9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int y = h + k / 3 - 1, x = w + k % 3 - 1;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
#pragma unroll
        for (int c = 0; c < 2; ++c) f[c][k] = in ? flow[((int64_t)b * 2 + c) * HW + (int64_t)y * W + x] * (float)S : 0.f;
    }
    const float *m = mask + (int64_t)b * 9 * S * S * HW + p;
    const int64_t OW = (int64_t)W * S;
#pragma unroll 1
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) {
            float e[9], mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                e[k] = m[(int64_t)(k * S * S + i * S + j) * HW];
                mx = fmaxf(mx, e[k]);
            }
            float sum = 0.f, a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                e[k] = expf(e[k] - mx);
                sum += e[k];
            }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float wk = e[k] * inv;
                a0 += wk * f[0][k];
                a1 += wk * f[1][k];
            }
            const int64_t o = ((int64_t)h * S + i) * OW + (int64_t)w * S + j;
            out[((int64_t)b * 2 + 0) * HW * S * S + o] = a0;
            out[((int64_t)b * 2 + 1) * HW * S * S + o] = a1;
        }
}

}  // namespace

RPE_API int rpe_convex_upsample(const float *flow, const float *mask, int B, int H, int W, int scale, float *out, rpe_stream_t stream) {
    if (!flow || !mask || !out || B < 0 || H < 1 || W < 1) return RPE_EINVAL;
    if (scale != 2 && scale != 4 && scale != 8) return RPE_EUNSUPPORTED;
    if (B == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    dim3 grid((unsigned)(((int64_t)H * W + 255) / 256), B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (scale == 2) hipLaunchKernelGGL(convex_upsample_kernel<2>, grid, block, 0, st, flow, mask, H, W, out);
    else if (scale == 4) hipLaunchKernelGGL(convex_upsample_kernel<4>, grid, block, 0, st, flow, mask, H, W, out);
    else hipLaunchKernelGGL(convex_upsample_kernel<8>, grid, block, 0, st, flow, mask, H, W, out);
    return rpe_launch_status();
}

RPE_API int rpe_dwconv3(const float *in0, int C0, const float *in1, int C1, const float *in2, int C2, const float *weight,
                        const float *bias, int B, int H, int W, int kh, int gate, float *out, rpe_stream_t stream) {
    const int C = C0 + C1 + C2;
    if (!in0 || !weight || !out || C0 < 1 || C1 < 0 || C2 < 0 || (C1 > 0 && !in1) || (C2 > 0 && !in2)) return RPE_EINVAL;
    if (B < 0 || H < 1 || W < 1 || (kh != 1 && kh != 3) || (kh == 1 && H != 1) || (gate && (C % 2))) return RPE_EINVAL;
    if (B == 0) return 0;
    if (B > 65535 || C > 65535) return RPE_EUNSUPPORTED;
    Segments seg{{in0, in1, in2}, {C0, C1, C2}};
    const int64_t HW = (int64_t)H * W;
    dim3 grid((unsigned)((HW + 255) / 256), gate ? C / 2 : C, B), block(256);
    hipStream_t st = (hipStream_t)stream;
    auto aligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (W % 4 == 0 && aligned(in0) && aligned(in1) && aligned(in2) && aligned(out)) {  // planes stay 16-byte aligned: HW % 4 == 0
        const int64_t strips = (int64_t)(W / 4) * (kh == 3 ? (H + 3) / 4 : 1);
        grid.x = (unsigned)((strips + 255) / 256);
        if (kh == 3) {
            if (gate) hipLaunchKernelGGL((dwconv3_strip_kernel<3, true>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
            else hipLaunchKernelGGL((dwconv3_strip_kernel<3, false>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
        } else {
            if (gate) hipLaunchKernelGGL((dwconv3_strip_kernel<1, true>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
            else hipLaunchKernelGGL((dwconv3_strip_kernel<1, false>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
        }
        return rpe_launch_status();
    }
    if (kh == 3) {
        if (gate) hipLaunchKernelGGL((dwconv3_kernel<3, true>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
        else hipLaunchKernelGGL((dwconv3_kernel<3, false>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
    } else {
        if (gate) hipLaunchKernelGGL((dwconv3_kernel<1, true>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
        else hipLaunchKernelGGL((dwconv3_kernel<1, false>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
    }
    return rpe_launch_status();
}

RPE_API int rpe_gdfn_tail(const float *t, int B, int hidden, int H, int W, int kh, const float *dw_weight, const float *dw_bias,
                          const float *packed_weight, int Cout, const float *shift, const float *residual, float *y, rpe_stream_t stream) {
    if (!t || !dw_weight || !packed_weight || !y || B < 0 || hidden < 1 || H < 1 || W < 1 || Cout < 1) return RPE_EINVAL;
    if ((kh != 1 && kh != 3) || (kh == 1 && H != 1)) return RPE_EINVAL;
    auto aligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (W % 4 != 0 || Cout > 128 || B > 65535 || !aligned(t) || !aligned(y) || !aligned(residual)) return RPE_EUNSUPPORTED;
    if (B == 0) return 0;
    const int64_t P = (int64_t)H * W;
    const int ktiles = (hidden + 3) / 4, n_otiles = (Cout + 15) / 16;
    dim3 grid((unsigned)((P + 63) / 64), (unsigned)B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (kh == 3) hipLaunchKernelGGL(gdfn_tail_kernel<3>, grid, block, 0, st, t, hidden, H, W, dw_weight, dw_bias, packed_weight, ktiles, n_otiles, Cout, shift, residual, y);
    else hipLaunchKernelGGL(gdfn_tail_kernel<1>, grid, block, 0, st, t, hidden, H, W, dw_weight, dw_bias, packed_weight, ktiles, n_otiles, Cout, shift, residual, y);
    return rpe_launch_status();
}

static int launch_layernorm(const NormJobs &jobs, int njobs, int B, int C, int64_t P, float eps, hipStream_t st) {
    if (B == 0 || P == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    if (C > 4 * 64) return RPE_EUNSUPPORTED;
    dim3 grid((unsigned)((P + 63) / 64), B, njobs), block(256);
    const int cpt = (C + 3) / 4;
#define RPE_LN(N) hipLaunchKernelGGL(channel_norm_kernel<N>, grid, block, 0, st, jobs, C, P, eps)
    if (cpt <= 8) RPE_LN(8);
    else if (cpt <= 16) RPE_LN(16);
    else if (cpt <= 24) RPE_LN(24);
    else if (cpt <= 32) RPE_LN(32);
    else if (cpt <= 48) RPE_LN(48);
    else RPE_LN(64);
#undef RPE_LN
    return rpe_launch_status();
}

RPE_API int rpe_channel_layernorm(const float *x0, const float *weight0, const float *bias0, float *out0, const float *x1,
                                  const float *weight1, const float *bias1, float *out1, int B, int C, int64_t P, float eps,
                                  rpe_stream_t stream) {
    if (!x0 || !weight0 || !out0 || B < 0 || C < 1 || P < 0) return RPE_EINVAL;
    if (!x1) {  // one map
        NormJobs jobs{{x0, x0}, {weight0, weight0}, {bias0, bias0}, {out0, out0}};
        return launch_layernorm(jobs, 1, B, C, P, eps, (hipStream_t)stream);
    }
    if (!weight1 || !out1) return RPE_EINVAL;
    NormJobs jobs{{x0, x1}, {weight0, weight1}, {bias0, bias1}, {out0, out1}};
    return launch_layernorm(jobs, 2, B, C, P, eps, (hipStream_t)stream);
}

RPE_API int rpe_channel_affine_act(float *y, const float *scale, const float *shift, int B, int C, int64_t P, int act,
                                   float slope, rpe_stream_t stream) {
    if (!y || B < 0 || C < 1 || P < 0 || act < 0 || act > 2) return RPE_EINVAL;
    if (B == 0 || P == 0) return 0;
    if (B > 65535 || C > 65535) return RPE_EUNSUPPORTED;
    dim3 grid((unsigned)((P + 1023) / 1024), C, B), block(256);
    hipLaunchKernelGGL(affine_act_kernel, grid, block, 0, (hipStream_t)stream, y, scale, shift, C, P, act, slope);
    return rpe_launch_status();
}

RPE_API int rpe_channel_affine_add_act(float *y, const float *scale, const float *shift, const float *z, const float *zscale, int B,
                                       int C, int64_t P, int act, float slope, rpe_stream_t stream) {
    if (!y || !z || B < 0 || C < 1 || P < 0 || act < 0 || act > 2) return RPE_EINVAL;
    if (B == 0 || P == 0) return 0;
    if (B > 65535 || C > 65535) return RPE_EUNSUPPORTED;
    dim3 grid((unsigned)((P + 1023) / 1024), C, B), block(256);
    hipLaunchKernelGGL(affine_add_act_kernel, grid, block, 0, (hipStream_t)stream, y, scale, shift, z, zscale, C, P, act, slope);
    return rpe_launch_status();
}

RPE_API int rpe_residual_tail(float *y, const float *scale, const float *shift, const float *x, const float *shortcut_weight,
                              const float *shortcut_scale, int B, int Cin, int Cout, int H, int W, int stride, int act, float slope,
                              rpe_stream_t stream) {
    if (!y || !x || !shortcut_weight || B < 0 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || stride < 1 || act < 0 || act > 2) return RPE_EINVAL;
    if (Cin > kTailMaxCin || B > 65535) return RPE_EUNSUPPORTED;
    if (B == 0) return 0;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;  // kernel 1, no padding
    dim3 grid((unsigned)((Ho * Wo + 255) / 256), (unsigned)((Cout + kTailOc - 1) / kTailOc), (unsigned)B), block(256);
    hipLaunchKernelGGL(residual_tail_kernel, grid, block, 0, (hipStream_t)stream, y, scale, shift, x, shortcut_weight, shortcut_scale, Cin, Cout,
                       H, W, Ho, Wo, stride, act, slope);
    return rpe_launch_status();
}
