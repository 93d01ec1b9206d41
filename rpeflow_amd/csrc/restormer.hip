// Memory-bound pieces of the Restormer cross-attention blocks (models/restormer_arch.py), SURVEY.md
// section 8(f) rank 1 -- the first widening beyond the hot path.  On the GPU the reference runs each of
// these as a chain of PyTorch kernels; MIOpen has no tuned depth-wise 3x3 fp32 solver for these shapes and
// falls back to naive_conv (65 us a call, 80 calls a forward).
//
//   dwconv3_kernel      depth-wise 3x3 (2-D) / 3-tap (1-D) convolution, stride 1, zero padding 1, optional bias
//                       (qkv_dwconv, restormer_arch.py:175-176, 256-257; dwconv :96-97, 234-235).  The input may be
//                       given as up to three channel segments, which fuses torch.cat((x, y, y)) (:182, 263);
//                       gate = 1 fuses the GDFN gate gelu(x1) * x2 over the two halves of the channels (:104-105, 244-245).
//   channel_norm_kernel LayerNorm over the channel axis of [B,C,P] (WithBias / BiasFree, :31-63): mean and biased
//                       variance per position in one pass (Welford), normalise + affine in a second.
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

__global__ __launch_bounds__(256) void channel_norm_kernel(const float *__restrict__ x, const float *__restrict__ weight,
                                                           const float *__restrict__ bias, int C, int64_t P, float eps,
                                                           float *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= P) return;
    const float *xb = x + (int64_t)b * C * P + p;
    float mean = 0.f, m2 = 0.f;
    for (int c = 0; c < C; ++c) {  // Welford: biased variance = m2 / C
        const float v = xb[(int64_t)c * P];
        const float d = v - mean;
        mean += d / (float)(c + 1);
        m2 += d * (v - mean);
    }
    const float inv = 1.0f / sqrtf(m2 / (float)C + eps);
    float *ob = out + (int64_t)b * C * P + p;
    if (bias) {
        for (int c = 0; c < C; ++c) ob[(int64_t)c * P] = (xb[(int64_t)c * P] - mean) * inv * weight[c] + bias[c];
    } else {
        for (int c = 0; c < C; ++c) ob[(int64_t)c * P] = xb[(int64_t)c * P] * inv * weight[c];
    }
}

}  // namespace

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
    if (kh == 3) {
        if (gate) hipLaunchKernelGGL((dwconv3_kernel<3, true>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
        else hipLaunchKernelGGL((dwconv3_kernel<3, false>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
    } else {
        if (gate) hipLaunchKernelGGL((dwconv3_kernel<1, true>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
        else hipLaunchKernelGGL((dwconv3_kernel<1, false>), grid, block, 0, st, seg, weight, bias, C, H, W, out);
    }
    return rpe_launch_status();
}

RPE_API int rpe_channel_layernorm(const float *x, const float *weight, const float *bias, int B, int C, int64_t P, float eps,
                                  float *out, rpe_stream_t stream) {
    if (!x || !weight || !out || B < 0 || C < 1 || P < 0) return RPE_EINVAL;
    if (B == 0 || P == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    dim3 grid((unsigned)((P + 255) / 256), B), block(256);
    hipLaunchKernelGGL(channel_norm_kernel, grid, block, 0, (hipStream_t)stream, x, weight, bias, C, P, eps, out);
    return rpe_launch_status();
}
