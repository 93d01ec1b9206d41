// Gradients of correlation2d with respect to both inputs (the training path of the operator:
// CorrelationFunction.backward, models/csrc/wrapper.py:27-37; reference kernels K2/K3,
// correlation_backward_kernel.cu:4-88, launch one workgroup of C threads per pixel and cap C at 1024).
//
//   out[b][k][p]      = 1/C sum_c in1[b][c][p] in2[b][c][p + d_k],   d_k = (k / n - md, k % n - md), zero outside the image
//   grad_in1[b][c][p] = 1/C sum_k go[b][k][p]       in2[b][c][p + d_k]
//   grad_in2[b][c][q] = 1/C sum_k go[b][k][q - d_k] in1[b][c][q - d_k]
//
// Both are one kernel: an (2md+1)^2-tap filter per pixel whose taps (the grad_out values, pre-shifted for grad_in2)
// sit in registers and are reused over all C channels; the feature plane tile + halo goes through LDS, CH channels per
// barrier.  fp32 VALU; this path is not on the inference hot path.
#include "common.h"

namespace {

constexpr int kTile = 16;  // 16 x 16 pixels per workgroup
constexpr int kCh = 8;     // channels staged per barrier

template <int MD, bool SECOND>
__global__ __launch_bounds__(kTile * kTile) void corr_backward_kernel(const float *__restrict__ go, const float *__restrict__ src,
                                                                      int C, int H, int W, float *__restrict__ grad) {
    constexpr int N = 2 * MD + 1, K = N * N, TW = kTile + 2 * MD;
    __shared__ float tile[kCh][TW][TW + 1];
    const int tx = threadIdx.x % kTile, ty = threadIdx.x / kTile;
    const int x0 = blockIdx.x * kTile, y0 = blockIdx.y * kTile, b = blockIdx.z;
    const int x = x0 + tx, y = y0 + ty;
    const bool live = x < W && y < H;
    const int64_t HW = (int64_t)H * W;
    const float inv_c = 1.0f / (float)C;

    // the taps of this pixel: go[k][p] (grad_in1) or go[k][p - d_k] (grad_in2), scaled by 1/C
    float g[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int dy = k / N - MD, dx = k % N - MD;
        const int gy = SECOND ? y - dy : y, gx = SECOND ? x - dx : x;
        const bool ok = live && gy >= 0 && gy < H && gx >= 0 && gx < W;
        g[k] = ok ? go[((int64_t)b * K + k) * HW + (int64_t)gy * W + gx] * inv_c : 0.f;
    }
    for (int c0 = 0; c0 < C; c0 += kCh) {
        __syncthreads();
        for (int e = threadIdx.x; e < kCh * TW * TW; e += kTile * kTile) {
            const int cc = e / (TW * TW), r = e % (TW * TW), ly = r / TW, lx = r % TW;
            const int sy = y0 + ly - MD, sx = x0 + lx - MD, c = c0 + cc;
            tile[cc][ly][lx] = (c < C && sy >= 0 && sy < H && sx >= 0 && sx < W) ? src[((int64_t)b * C + c) * HW + (int64_t)sy * W + sx] : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int cc = 0; cc < kCh && c0 + cc < C; ++cc) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int dy = k / N - MD, dx = k % N - MD;
                // grad_in1 reads the source at p + d_k, grad_in2 at q - d_k
                acc = __fmaf_rn(g[k], tile[cc][ty + MD + (SECOND ? -dy : dy)][tx + MD + (SECOND ? -dx : dx)], acc);
            }
            if (live) grad[((int64_t)b * C + c0 + cc) * HW + (int64_t)y * W + x] = acc;
        }
    }
}

template <int MD>
void launch_both(const float *go, const float *in1, const float *in2, int B, int C, int H, int W, float *g1, float *g2, hipStream_t st) {
    dim3 grid((W + kTile - 1) / kTile, (H + kTile - 1) / kTile, B), block(kTile * kTile);
    if (g1) hipLaunchKernelGGL((corr_backward_kernel<MD, false>), grid, block, 0, st, go, in2, C, H, W, g1);
    if (g2) hipLaunchKernelGGL((corr_backward_kernel<MD, true>), grid, block, 0, st, go, in1, C, H, W, g2);
}

}  // namespace

RPE_API int rpe_correlation2d_backward(const float *grad_out, const float *in1, const float *in2, int B, int C, int H, int W, int md,
                                       float *grad_in1, float *grad_in2, rpe_stream_t stream) {
    if (!grad_out || !in1 || !in2 || (!grad_in1 && !grad_in2) || B < 0 || C < 1 || H < 1 || W < 1 || md < 0) return RPE_EINVAL;
    if (B == 0) return 0;
    if (B > 65535 || md > 4) return RPE_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    switch (md) {
        case 0: launch_both<0>(grad_out, in1, in2, B, C, H, W, grad_in1, grad_in2, st); break;
        case 1: launch_both<1>(grad_out, in1, in2, B, C, H, W, grad_in1, grad_in2, st); break;
        case 2: launch_both<2>(grad_out, in1, in2, B, C, H, W, grad_in1, grad_in2, st); break;
        case 3: launch_both<3>(grad_out, in1, in2, B, C, H, W, grad_in1, grad_in2, st); break;
        default: launch_both<4>(grad_out, in1, in2, B, C, H, W, grad_in1, grad_in2, st); break;
    }
    return rpe_launch_status();
}
