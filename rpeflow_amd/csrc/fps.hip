// furthest_point_sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel<1024> (furthest_point_sampling_kernel.cu:34-85:
// running distances in a global scratch buffer, ~11 barriers per sample) with one
// 1024-thread workgroup per cloud that keeps its points AND their running minimum
// distances in registers (PPT points per thread), a copy of the coordinates in LDS
// for the "look up the winner" step, wave butterflies for the argmax and ONE barrier
// per sample (the per-wave partials are double-buffered by sample parity).
//
// Rules are the CPU fallback's (wrapper.py:83-96), not the CUDA kernel's: start at
// index 0, nd = fl(fl(dx*dx + dy*dy) + dz*dz) with every square rounded, running
// min, next sample = FIRST index holding the maximum.
#include <math.h>

#include "common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / RPE_WAVE;  // 16

__device__ __forceinline__ void argmax_merge(float &v, int &i, float ov, int oi) {
    const bool take = (ov > v) || (ov == v && oi < i);
    v = take ? ov : v;
    i = take ? oi : i;
}

template <int PPT, bool LDS_XYZ>
__global__ __launch_bounds__(kThreads) void fps_kernel(const float *__restrict__ xyz, int64_t sb, int64_t sn, int64_t sd,
                                                       int N, int S, int64_t *__restrict__ idx) {
    extern __shared__ float lds[];
    // layout: [2][kWaves] partial values | [2][kWaves] partial indices | x[N] y[N] z[N]
    float *part_v = lds;
    int *part_i = reinterpret_cast<int *>(lds + 2 * kWaves);
    float *lx = lds + 4 * kWaves, *ly = lx + N, *lz = ly + N;

    const int tid = threadIdx.x, lane = rpe_lane(), wave = tid >> 6;
    const int b = blockIdx.x;
    xyz += (int64_t)b * sb;
    idx += (int64_t)b * S;

    float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int i = tid + j * kThreads;
        const bool valid = i < N;
        const float *a = xyz + (int64_t)(valid ? i : 0) * sn;
        px[j] = a[0];
        py[j] = a[sd];
        pz[j] = a[2 * sd];
        md[j] = valid ? 1e10f : -INFINITY;  // -inf: can never win the argmax, min() keeps it
        if (LDS_XYZ && valid) {
            lx[i] = px[j];
            ly[i] = py[j];
            lz[i] = pz[j];
        }
    }
    __syncthreads();

    int cur = 0;
    for (int s = 0; s < S; ++s) {
        if (tid == 0) idx[s] = (int64_t)cur;
        if (s == S - 1) break;
        float cx, cy, cz;
        if (LDS_XYZ) {
            cx = lx[cur]; cy = ly[cur]; cz = lz[cur];
        } else {
            const float *a = xyz + (int64_t)cur * sn;
            cx = a[0]; cy = a[sd]; cz = a[2 * sd];
        }
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
            const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
            float nd = xx + yy;
            nd = nd + zz;
            md[j] = nd < md[j] ? nd : md[j];
            const bool take = md[j] > bv;  // strict: first index wins inside a thread
            bv = take ? md[j] : bv;
            bi = take ? tid + j * kThreads : bi;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) argmax_merge(bv, bi, __shfl_xor(bv, off), __shfl_xor(bi, off));
        const int par = (s & 1) * kWaves;
        if (lane == 0) {
            part_v[par + wave] = bv;
            part_i[par + wave] = bi;
        }
        __syncthreads();
        float wv = part_v[par + (lane & (kWaves - 1))];
        int wi = part_i[par + (lane & (kWaves - 1))];
#pragma unroll
        for (int off = kWaves / 2; off > 0; off >>= 1) argmax_merge(wv, wi, __shfl_xor(wv, off), __shfl_xor(wi, off));
        cur = rpe_uniform(wi);
    }
}

template <int PPT>
int launch_fps(const float *xyz, int64_t sb, int64_t sn, int64_t sd, int B, int N, int S, int64_t *idx, hipStream_t st) {
    const size_t small = sizeof(float) * 4 * kWaves;
    const size_t full = small + sizeof(float) * 3 * (size_t)N;
    const bool use_lds = full <= 150 * 1024;
    const size_t shmem = use_lds ? full : small;
    auto kern = use_lds ? fps_kernel<PPT, true> : fps_kernel<PPT, false>;
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(kThreads), shmem, st, xyz, sb, sn, sd, N, S, idx);
    return rpe_launch_status();
}

}  // namespace

RPE_API int rpe_fps(const float *xyz, int64_t sb, int64_t sn, int64_t sd, int B, int N, int S, int64_t *idx,
                    rpe_stream_t stream) {
    if (!xyz || !idx || B < 0 || N <= 0 || S < 0 || S > N) return RPE_EINVAL;
    if (B == 0 || S == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int ppt = (N + kThreads - 1) / kThreads;
    if (ppt <= 1) return launch_fps<1>(xyz, sb, sn, sd, B, N, S, idx, st);
    if (ppt <= 2) return launch_fps<2>(xyz, sb, sn, sd, B, N, S, idx, st);
    if (ppt <= 4) return launch_fps<4>(xyz, sb, sn, sd, B, N, S, idx, st);
    if (ppt <= 8) return launch_fps<8>(xyz, sb, sn, sd, B, N, S, idx, st);
    if (ppt <= 16) return launch_fps<16>(xyz, sb, sn, sd, B, N, S, idx, st);
    if (ppt <= 32) return launch_fps<32>(xyz, sb, sn, sd, B, N, S, idx, st);
    return RPE_EUNSUPPORTED;
}
