// Evaluation metric sums on the device: eval_withocc.py:65-108 / eval_noocc.py:57-99.
//
// The reference walks the batch sample by sample and pulls ~12 scalars to the host with .item(); a tensor-op restatement
// is ~45 small launches over the 544 x 960 maps per batch (0.6 ms between two 17.7 ms forwards).  Here: ONE pass over
// the 2-D maps and the 3-D points (per-block partial sums in float64), then a one-block kernel that adds the partials
// to the twelve accumulators in a fixed order -- no atomics, results repeat bit for bit.
//
// Per element, in the reference's fp32 arithmetic (-ffp-contract=off: every operation rounded where the source says):
//   epe  = sqrt(sum_c (pred_c - gt_c)^2)            torch.sqrt(torch.sum(diff ** 2, dim=0)), channels left to right
//   mask = (gt_mask > 0 if the target carries a mask channel) and not isnan(epe)
//   2-D:  count, sum epe, #(epe < 1), #(epe > 3 and epe / sqrt(sum_c gt_c^2) > 0.05)          (eval_withocc.py:71-90)
//   3-D:  count, sum epe, #(epe < 0.05), #(epe < 0.1); the same again over mask and occ == 0  (:92-108)
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kSums = 12;
constexpr int64_t kMaxBlocks2 = 512, kMaxBlocks3 = 128;  // the one-block reduce kernel adds the partials of every block in order

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// block-level sums of NS doubles -> out[NS] (thread 0 writes); fixed order: lanes by butterfly, waves 0..3 left to right
template <int NS>
__device__ __forceinline__ void block_sums(double (&v)[NS], double *out) {
    __shared__ double part[kThreads / RPE_WAVE][NS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double s = wave_sum(v[i]);
        if (lane == 0) part[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / RPE_WAVE; ++w) s += part[w][threadIdx.x];
        out[threadIdx.x] = s;
    }
}

// blocks [0, g2): pixels; blocks [g2, g2 + g3): points.  partial[block][12] (entries a block does not own: zero)
__global__ __launch_bounds__(kThreads) void eval_partial_kernel(const float *__restrict__ p2, const float *__restrict__ t2, int c2, int64_t hw,
                                                                int64_t n2, const float *__restrict__ p3, const float *__restrict__ t3, int c3,
                                                                int64_t np, int64_t n3, const float *__restrict__ occ, int g2,
                                                                double *__restrict__ partial) {
    double *out = partial + (int64_t)blockIdx.x * kSums;
    if ((int)blockIdx.x < g2) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n2; i += (int64_t)g2 * kThreads) {
            const int64_t b = i / hw, p = i - b * hw;
            const float *pr = p2 + b * 2 * hw + p, *gt = t2 + b * c2 * hw + p;
            const float gx = gt[0], gy = gt[hw];
            const float dx = pr[0] - gx, dy = pr[hw] - gy;
            const float sq = dx * dx + dy * dy;
            const float epe = __fsqrt_rn(sq);
            bool m = c2 > 2 ? gt[2 * hw] > 0.f : true;
            m = m && !(epe != epe);
            const float gn = __fsqrt_rn(gx * gx + gy * gy);
            const bool fl = epe > 3.0f && __fdiv_rn(epe, gn) > 0.05f;
            if (m) {
                s[0] += 1.0;
                s[1] += (double)epe;
                s[2] += epe < 1.0f ? 1.0 : 0.0;
                s[3] += fl ? 1.0 : 0.0;
            }
        }
        block_sums<4>(s, out);
        if (threadIdx.x >= 4 && threadIdx.x < kSums) out[threadIdx.x] = 0.0;
    } else {
        const int blk = blockIdx.x - g2, g3 = gridDim.x - g2;
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int64_t i = (int64_t)blk * kThreads + threadIdx.x; i < n3; i += (int64_t)g3 * kThreads) {
            const int64_t b = i / np, p = i - b * np;
            const float *pr = p3 + b * 3 * np + p, *gt = t3 + b * c3 * np + p;
            const float dx = pr[0] - gt[0], dy = pr[np] - gt[np], dz = pr[2 * np] - gt[2 * np];
            const float sq = (dx * dx + dy * dy) + dz * dz;
            const float epe = __fsqrt_rn(sq);
            bool m = c3 > 3 ? gt[3 * np] > 0.f : true;
            m = m && !(epe != epe);
            const double a5 = epe < 0.05f ? 1.0 : 0.0, a10 = epe < 0.1f ? 1.0 : 0.0;
            if (m) {
                s[0] += 1.0;
                s[1] += (double)epe;
                s[2] += a5;
                s[3] += a10;
                if (occ && occ[i] == 0.f) {
                    s[4] += 1.0;
                    s[5] += (double)epe;
                    s[6] += a5;
                    s[7] += a10;
                }
            }
        }
        __shared__ double tmp[8];
        block_sums<8>(s, tmp);
        __syncthreads();
        if (threadIdx.x < kSums) out[threadIdx.x] = threadIdx.x < 4 ? 0.0 : tmp[threadIdx.x - 4];
    }
}

// acc[j] += sum over blocks of partial[block][j] in a fixed order: 16 lanes per sum take every 16th block, then a butterfly
__global__ __launch_bounds__(kSums * 16) void eval_reduce_kernel(const double *__restrict__ partial, int blocks, double *__restrict__ acc) {
    const int j = threadIdx.x >> 4, l = threadIdx.x & 15;
    double s = 0.0;
    for (int b = l; b < blocks; b += 16) s += partial[(int64_t)b * kSums + j];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (l == 0) acc[j] += s;
}

}  // namespace

RPE_API int rpe_eval_workspace_doubles(int64_t n_pixels, int64_t n_points) {
    const int64_t g2 = (n_pixels + 4 * kThreads - 1) / (4 * kThreads), g3 = (n_points + kThreads - 1) / kThreads;
    const int64_t blocks = (g2 < kMaxBlocks2 ? (g2 > 0 ? g2 : 1) : kMaxBlocks2) + (g3 < kMaxBlocks3 ? (g3 > 0 ? g3 : 1) : kMaxBlocks3);
    return (int)(blocks * kSums);
}

RPE_API int rpe_eval_accumulate(const float *flow2d, const float *target2d, int target2d_channels, int B, int64_t HW,
                                const float *flow3d, const float *target3d, int target3d_channels, int64_t N, const float *occ_mask,
                                double *workspace, double *acc, rpe_stream_t stream) {
    if (!flow2d || !target2d || !flow3d || !target3d || !workspace || !acc) return RPE_EINVAL;
    if (B < 0 || HW < 0 || N < 0 || target2d_channels < 2 || target2d_channels > 3 || target3d_channels < 3 || target3d_channels > 4)
        return RPE_EINVAL;
    if (B == 0) return 0;
    const int64_t n2 = (int64_t)B * HW, n3 = (int64_t)B * N;
    int64_t g2 = (n2 + 4 * kThreads - 1) / (4 * kThreads), g3 = (n3 + kThreads - 1) / kThreads;
    g2 = g2 < 1 ? 1 : g2 > kMaxBlocks2 ? kMaxBlocks2 : g2;
    g3 = g3 < 1 ? 1 : g3 > kMaxBlocks3 ? kMaxBlocks3 : g3;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(eval_partial_kernel, dim3((unsigned)(g2 + g3)), dim3(kThreads), 0, st, flow2d, target2d, target2d_channels, HW, n2,
                       flow3d, target3d, target3d_channels, N, n3, occ_mask, (int)g2, workspace);
    hipLaunchKernelGGL(eval_reduce_kernel, dim3(1), dim3(kSums * 16), 0, st, workspace, (int)(g2 + g3), acc);
    return rpe_launch_status();
}
