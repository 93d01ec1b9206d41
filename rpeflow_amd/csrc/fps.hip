// furthest_point_sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel<1024> (furthest_point_sampling_kernel.cu:34-85:
// running distances in a global scratch buffer, ~11 barriers per sample) with one
// 1024-thread workgroup per cloud that keeps its points AND their running minimum
// distances in registers (PPT points per thread) and needs ONE barrier per sample
// (the per-wave partials are double-buffered by sample parity).  Two kernels:
//   fps_kernel_int      every point recomputed per sample; integer-pipe running distances, fused DPP reductions
//   fps_pruned2_kernel  Morton-sorted clusters, a wave whose box the new sample cannot reach skips its recompute
//                       (exact: a skipped point provably keeps its distance); the default for long runs on big clouds
// Both give identical indices (the GPU tests cross-check them against each other and the CPU restatement).
//
// Rules are the CPU fallback's (wrapper.py:83-96), not the CUDA kernel's: start at
// index 0, nd = fl(fl(dx*dx + dy*dy) + dz*dz) with every square rounded, running
// min, next sample = FIRST index holding the maximum.
#include <math.h>

#include "common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / RPE_WAVE;  // 16


// ---- DPP reductions (gfx9 row_shr / row_bcast controls): ~8 cycles a step instead of a
// ds_bpermute round trip per __shfl_xor.  After the six steps lane 63 holds the result;
// after the first four, lane 15 of every row holds its row's result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max(float v) {
    const int o = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return fmaxf(v, __int_as_float(o));
}
__device__ __forceinline__ float row_max16(float v) {
    v = dpp_max<0x111, 0xf>(v);  // row_shr:1
    v = dpp_max<0x112, 0xf>(v);  // row_shr:2
    v = dpp_max<0x114, 0xf>(v);  // row_shr:4
    v = dpp_max<0x118, 0xf>(v);  // row_shr:8
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = row_max16(v);
    v = dpp_max<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
    v = dpp_max<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3
    return rpe_readlane(v, 63);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- fps_kernel_int: running distances on the integer pipe, reductions as fused DPP instructions --------------------
// Running distances are non-negative floats (sums of squares, 1e10 at the start), so their bit patterns order like
// the values: min / max / == run as v_min_i32 / v_max3_i32 / v_cmp_eq_u32 with no NaN canonicalisation, invalid slots
// hold -1.  The compiler expands every __builtin_amdgcn_update_dpp + max step into mov, nop, mov_dpp, 2 x max; here a
// step is ONE v_max_i32_dpp (plus the two wait states a DPP read of a fresh VALU result needs).  The argmax keeps no
// per-point index: a wave reduces values only, then eight v_cmp ballots find the lowest index holding the wave's
// maximum on the scalar unit.  Per sample and wave ~85 VALU instructions instead of ~160; tie rule unchanged.
#define RPE_DPP_STEP(op, v, ctrl) asm volatile("s_nop 1\n\t" op " %0, %0, %0 " ctrl : "+v"(v))
__device__ __forceinline__ int row_max16_i32(int v) {
    RPE_DPP_STEP("v_max_i32_dpp", v, "row_shr:1 row_mask:0xf bank_mask:0xf");
    RPE_DPP_STEP("v_max_i32_dpp", v, "row_shr:2 row_mask:0xf bank_mask:0xf");
    RPE_DPP_STEP("v_max_i32_dpp", v, "row_shr:4 row_mask:0xf bank_mask:0xf");
    RPE_DPP_STEP("v_max_i32_dpp", v, "row_shr:8 row_mask:0xf bank_mask:0xf");
    return v;  // lane 15 of every row: the row's maximum
}
__device__ __forceinline__ int wave_max_i32(int v) {
    v = row_max16_i32(v);
    RPE_DPP_STEP("v_max_i32_dpp", v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    RPE_DPP_STEP("v_max_i32_dpp", v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    int r;
    asm volatile("s_nop 1\n\tv_readlane_b32 %0, %1, 63" : "=s"(r) : "v"(v));
    return r;
}

template <int PPT, bool LDS_XYZ>
__global__ __launch_bounds__(kThreads) void fps_kernel_int(const float *__restrict__ xyz, int64_t sb, int64_t sn, int64_t sd,
                                                           int N, int S, int64_t *__restrict__ idx) {
    static_assert(PPT % 2 == 0, "packed path needs an even number of points per thread");
    constexpr int H = PPT / 2;
    extern __shared__ float lds[];
    int *part_v = reinterpret_cast<int *>(lds);
    int *part_i = reinterpret_cast<int *>(lds + 2 * kWaves);
    float *lx = lds + 4 * kWaves, *ly = lx + N, *lz = ly + N;

    const int tid = threadIdx.x, lane = rpe_lane();
    const int wave = rpe_uniform(tid >> 6);
    const int b = blockIdx.x;
    xyz += (int64_t)b * sb;
    idx += (int64_t)b * S;

    f32x2 px[H], py[H], pz[H];
    int md[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int i = tid + j * kThreads;
        const bool valid = i < N;
        const float *a = xyz + (int64_t)(valid ? i : 0) * sn;
        const float x = a[0], y = a[sd], z = a[2 * sd];
        px[j >> 1][j & 1] = x;
        py[j >> 1][j & 1] = y;
        pz[j >> 1][j & 1] = z;
        md[j] = valid ? __float_as_int(1e10f) : -1;
        if (LDS_XYZ && valid) {
            lx[i] = x;
            ly[i] = y;
            lz[i] = z;
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
        const f32x2 cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
        int tmax = -1;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const f32x2 dx = px[h] - cx2, dy = py[h] - cy2, dz = pz[h] - cz2;
            const f32x2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
            f32x2 nd = xx + yy;
            nd = nd + zz;
            md[2 * h] = min(md[2 * h], __float_as_int(nd[0]));
            md[2 * h + 1] = min(md[2 * h + 1], __float_as_int(nd[1]));
            tmax = max(tmax, max(md[2 * h], md[2 * h + 1]));
        }
        const int wmax = wave_max_i32(tmax);
        // lowest index in the wave holding wmax: index = tid + j * kThreads, so the lowest j wins, then the lowest lane
        int widx = 0x7fffffff;
#pragma unroll
        for (int j = PPT - 1; j >= 0; --j) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(md[j] == wmax);
            widx = m ? j * kThreads + wave * RPE_WAVE + (int)__builtin_ctzll(m) : widx;
        }
        const int par = (s & 1) * kWaves;
        if (lane == 0) {
            part_v[par + wave] = wmax;
            part_i[par + wave] = widx;
        }
        __syncthreads();
        const int pv = part_v[par + (lane & (kWaves - 1))];
        const int pi = part_i[par + (lane & (kWaves - 1))];
        int bmax;
        {
            int r = row_max16_i32(pv);
            asm volatile("s_nop 1\n\tv_readlane_b32 %0, %1, 15" : "=s"(bmax) : "v"(r));
        }
        unsigned long long tied = __builtin_amdgcn_ballot_w64(pv == bmax) & 0xffffull;  // row 0 holds one copy of the 16 partials
        cur = 0x7fffffff;
        do {  // one pass unless two waves tie on the maximum
            cur = min(cur, __builtin_amdgcn_readlane(pi, (int)__builtin_ctzll(tied)));
            tied &= tied - 1;
        } while (tied);
    }
}


__device__ __forceinline__ unsigned spread10(unsigned v) {  // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_minf(float v) {
    const int o = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return fminf(v, __int_as_float(o));
}
__device__ __forceinline__ float wave_minf(float v) {
    v = dpp_minf<0x111, 0xf>(v); v = dpp_minf<0x112, 0xf>(v); v = dpp_minf<0x114, 0xf>(v); v = dpp_minf<0x118, 0xf>(v);
    v = dpp_minf<0x142, 0xa>(v); v = dpp_minf<0x143, 0xc>(v);
    return rpe_readlane(v, 63);
}

// ---- Morton-sorted clusters, skipped per wave ---------------------------------------------------------------------
// A new sample c lowers the running distance of point i only if d(i,c) < md[i].  Points are sorted along a Morton
// curve so that a wave owns a spatially compact cluster (PPT rows of 64 points) with a known bounding box.  Per sample
// a wave evaluates a lower bound LB of the squared distance from c to its box; if 0.99999 * LB exceeds the cluster's
// largest running distance no point of the cluster can change (the margin is far larger than the rounding of either
// side) and the cluster is skipped: no distance arithmetic, no reduction.  Late in the run a few of the 16 waves do any
// work per sample.  Whatever is recomputed uses exactly the reference arithmetic, ties resolve to the lowest ORIGINAL
// index, so the output is index-for-index the reference fallback's.
__device__ __forceinline__ int wave_min_i32(int v) {
    RPE_DPP_STEP("v_min_i32_dpp", v, "row_shr:1 row_mask:0xf bank_mask:0xf");
    RPE_DPP_STEP("v_min_i32_dpp", v, "row_shr:2 row_mask:0xf bank_mask:0xf");
    RPE_DPP_STEP("v_min_i32_dpp", v, "row_shr:4 row_mask:0xf bank_mask:0xf");
    RPE_DPP_STEP("v_min_i32_dpp", v, "row_shr:8 row_mask:0xf bank_mask:0xf");
    RPE_DPP_STEP("v_min_i32_dpp", v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    RPE_DPP_STEP("v_min_i32_dpp", v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    int r;
    asm volatile("s_nop 1\n\tv_readlane_b32 %0, %1, 63" : "=s"(r) : "v"(v));
    return r;
}

#ifdef RPE_FPS_STATS
__device__ unsigned long long g_fps_stats[4];
#endif
// ---- fps_pruned2_kernel: the skip test folded into the post-barrier reduction ---------------------------------------
// One Morton cluster per wave; (a) every wave publishes its candidate's COORDINATES with its
// partial, so the winner's coordinates are three v_readlane away instead of an LDS lookup, and (b) the skip test is
// evaluated for all 16 candidates at once -- lane l tests candidate l against this wave's box -- while the DPP
// reduction that picks the winner is in flight; the winner's lane then selects its bit.  A skipping wave's iteration
// is: read 16 partials, reduce, readlane, republish.  The start needs no special case: every valid point begins at
// 1e10, all waves tie, the lowest original index (0) wins.
template <int PPT>
__global__ __launch_bounds__(kThreads) void fps_pruned2_kernel(const float *__restrict__ xyz, int64_t sb, int64_t sn, int64_t sd,
                                                               int N, int S, int64_t *__restrict__ idx) {
    static_assert(PPT % 2 == 0, "packed path needs an even number of points per thread");
    constexpr int NP = PPT * kThreads;
    constexpr int H = PPT / 2;
    extern __shared__ unsigned long long sortbuf[];
    __shared__ float red[6][kWaves];
    __shared__ int part[2][5][kWaves];  // [parity][value bits, index, x, y, z][wave]
    const int tid = threadIdx.x, lane = rpe_lane();
    const int wave = rpe_uniform(tid >> 6);
    const int b = blockIdx.x;
    xyz += (int64_t)b * sb;
    idx += (int64_t)b * S;

    // ---- 1. bounding box of the cloud, 2. Morton keys + bitonic sort (as fps_pruned_kernel)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < N; i += kThreads) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = xyz[(int64_t)i * sn + d * sd];
            lo[d] = fminf(lo[d], v);
            hi[d] = fmaxf(hi[d], v);
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float l = wave_minf(lo[d]), h = wave_max(hi[d]);
        if (lane == 0) { red[d][wave] = l; red[3 + d][wave] = h; }
    }
    __syncthreads();
    float clo[3], scale[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float l = red[d][0], h = red[3 + d][0];
        for (int w = 1; w < kWaves; ++w) { l = fminf(l, red[d][w]); h = fmaxf(h, red[3 + d][w]); }
        clo[d] = l;
        scale[d] = h > l ? 1023.0f / (h - l) : 0.f;
    }
    for (int i = tid; i < NP; i += kThreads) {
        unsigned long long e = ~0ull;
        if (i < N) {
            unsigned key = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float v = xyz[(int64_t)i * sn + d * sd];
                int q = (int)((v - clo[d]) * scale[d]);
                q = q < 0 ? 0 : (q > 1023 ? 1023 : q);
                key |= spread10((unsigned)q) << d;
            }
            e = ((unsigned long long)key << 32) | (unsigned)i;
        }
        sortbuf[i] = e;
    }
    __syncthreads();
    for (int k = 2; k <= NP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = tid; p < NP / 2; p += kThreads) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
                const unsigned long long a = sortbuf[i], c = sortbuf[l];
                const bool up = (i & k) == 0;
                if ((a > c) == up) { sortbuf[i] = c; sortbuf[l] = a; }
            }
            __syncthreads();
        }
    }

    // ---- 3. this thread's points (row j of wave w = sorted positions w*64*PPT + j*64 + lane) and the wave's box
    f32x2 px[H], py[H], pz[H];
    int md[PPT], oi[PPT];
    float l3[3] = {INFINITY, INFINITY, INFINITY}, h3[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const unsigned o = (unsigned)(sortbuf[wave * (RPE_WAVE * PPT) + j * RPE_WAVE + lane] & 0xffffffffu);
        const bool valid = o != 0xffffffffu;
        oi[j] = valid ? (int)o : 0x7fffffff;
        const float *a = xyz + (int64_t)(valid ? o : 0) * sn;
        const float x = a[0], y = a[sd], z = a[2 * sd];
        px[j >> 1][j & 1] = x;
        py[j >> 1][j & 1] = y;
        pz[j >> 1][j & 1] = z;
        md[j] = valid ? __float_as_int(1e10f) : -1;
        if (valid) {
            l3[0] = fminf(l3[0], x); h3[0] = fmaxf(h3[0], x);
            l3[1] = fminf(l3[1], y); h3[1] = fmaxf(h3[1], y);
            l3[2] = fminf(l3[2], z); h3[2] = fmaxf(h3[2], z);
        }
    }
    float blo[3], bhi[3];  // wave-uniform
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        blo[d] = wave_minf(l3[d]);
        bhi[d] = wave_max(h3[d]);
    }

    // the wave's candidate (largest running distance, lowest original index holding it, that point's coordinates)
    int wmax, widx;
    float wx, wy, wz;
    auto candidate = [&](int tmax) {
        wmax = wave_max_i32(tmax);
        unsigned long long m[PPT];
        int holders = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            m[j] = __builtin_amdgcn_ballot_w64(md[j] == wmax);
            holders += (int)__builtin_popcountll(m[j]);
        }
        if (holders != 1) {  // several points share the maximum (the start, duplicates): lowest original index among them
            int v = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < PPT; ++j) v = md[j] == wmax ? min(v, oi[j]) : v;
            const int best = wave_min_i32(v);
#pragma unroll
            for (int j = 0; j < PPT; ++j) m[j] = __builtin_amdgcn_ballot_w64(md[j] == wmax && oi[j] == best);
        }
        widx = 0x7fffffff;
        wx = wy = wz = 0.f;
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if (m[j]) {  // wave-uniform; exactly one j has a (single) bit set unless the wave is empty
                const int l = (int)__builtin_ctzll(m[j]);
                widx = __builtin_amdgcn_readlane(oi[j], l);
                wx = rpe_readlane(px[j >> 1][j & 1], l);
                wy = rpe_readlane(py[j >> 1][j & 1], l);
                wz = rpe_readlane(pz[j >> 1][j & 1], l);
            }
    };
    {
        int t = -1;
#pragma unroll
        for (int j = 0; j < PPT; ++j) t = max(t, md[j]);
        candidate(t);
    }
    __syncthreads();  // the sort buffer is dead; part[] is a separate array

    for (int s = 0;; ++s) {
        const int par = s & 1;
        if (lane == 0) {
            part[par][0][wave] = wmax;
            part[par][1][wave] = widx;
            part[par][2][wave] = __float_as_int(wx);
            part[par][3][wave] = __float_as_int(wy);
            part[par][4][wave] = __float_as_int(wz);
        }
        __syncthreads();
        const int l16 = lane & (kWaves - 1);
        const int pv = part[par][0][l16], pi = part[par][1][l16];
        const float qx = __int_as_float(part[par][2][l16]), qy = __int_as_float(part[par][3][l16]), qz = __int_as_float(part[par][4][l16]);
        // lane l: would candidate l, as the next sample, lower any running distance of THIS wave's points?
        const float ex = fmaxf(fmaxf(blo[0] - qx, qx - bhi[0]), 0.f), ey = fmaxf(fmaxf(blo[1] - qy, qy - bhi[1]), 0.f),
                    ez = fmaxf(fmaxf(blo[2] - qz, qz - bhi[2]), 0.f);
        const float lb = ((ex * ex + ey * ey) + ez * ez) * 0.99999f;
        const unsigned long long needmask = __builtin_amdgcn_ballot_w64(__float_as_int(lb) <= wmax);
        int bmax;
        {
            int r = row_max16_i32(pv);
            asm volatile("s_nop 1\n\tv_readlane_b32 %0, %1, 15" : "=s"(bmax) : "v"(r));
        }
        unsigned long long tied = __builtin_amdgcn_ballot_w64(pv == bmax) & 0xffffull;
        int win = (int)__builtin_ctzll(tied), cur = __builtin_amdgcn_readlane(pi, win);
        for (tied &= tied - 1; tied; tied &= tied - 1) {  // two waves tie on the maximum: lowest original index
            const int l = (int)__builtin_ctzll(tied), c = __builtin_amdgcn_readlane(pi, l);
            if (c < cur) { cur = c; win = l; }
        }
        if (tid == 0) idx[s] = (int64_t)cur;
        if (s == S - 1) break;
        if ((needmask >> win) & 1ull) {  // wave-uniform
            const float cx = rpe_readlane(qx, win), cy = rpe_readlane(qy, win), cz = rpe_readlane(qz, win);
            const f32x2 cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
            int t = -1;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const f32x2 dx = px[h] - cx2, dy = py[h] - cy2, dz = pz[h] - cz2;
                const f32x2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
                f32x2 nd = xx + yy;
                nd = nd + zz;
                md[2 * h] = min(md[2 * h], __float_as_int(nd[0]));
                md[2 * h + 1] = min(md[2 * h + 1], __float_as_int(nd[1]));
                t = max(t, max(md[2 * h], md[2 * h + 1]));
            }
            candidate(t);
        }
    }
}

template <int PPT>
int launch_fps_pruned2(const float *xyz, int64_t sb, int64_t sn, int64_t sd, int B, int N, int S, int64_t *idx, hipStream_t st) {
    const size_t shmem = sizeof(unsigned long long) * (size_t)PPT * kThreads;
    auto kern = fps_pruned2_kernel<PPT>;
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(kThreads), shmem, st, xyz, sb, sn, sd, N, S, idx);
    return rpe_launch_status();
}

template <int PPT>
int launch_fps_plain(const float *xyz, int64_t sb, int64_t sn, int64_t sd, int B, int N, int S, int64_t *idx, hipStream_t st) {
    const size_t small = sizeof(float) * 4 * kWaves;
    const size_t full = small + sizeof(float) * 3 * (size_t)N;
    const bool use_lds = full <= 150 * 1024;
    const size_t shmem = use_lds ? full : small;
    auto kern = use_lds ? fps_kernel_int<PPT, true> : fps_kernel_int<PPT, false>;
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(kThreads), shmem, st, xyz, sb, sn, sd, N, S, idx);
    return rpe_launch_status();
}

}  // namespace

RPE_API int rpe_fps_algo(const float *xyz, int64_t sb, int64_t sn, int64_t sd, int B, int N, int S, int64_t *idx, int algo,
                         rpe_stream_t stream) {
    if (!xyz || !idx || B < 0 || N <= 0 || S < 0 || S > N) return RPE_EINVAL;
    if (algo != RPE_FPS_AUTO && algo != RPE_FPS_PLAIN && algo != RPE_FPS_PRUNED) return RPE_EINVAL;
    if (B == 0 || S == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int ppt = (N + kThreads - 1) / kThreads;
    const bool can_prune = N > kThreads && N <= 16 * kThreads;
    if (algo == RPE_FPS_PRUNED && !can_prune) return RPE_EUNSUPPORTED;
    // auto: the Morton sort of the pruned kernels costs ~100 us per launch; it pays from a few thousand samples on
    const bool auto_pruned = algo == RPE_FPS_AUTO && can_prune && N >= 8 * kThreads && S >= 2048;
    if (algo == RPE_FPS_PRUNED || auto_pruned) {
        if (ppt <= 2) return launch_fps_pruned2<2>(xyz, sb, sn, sd, B, N, S, idx, st);
        if (ppt <= 4) return launch_fps_pruned2<4>(xyz, sb, sn, sd, B, N, S, idx, st);
        if (ppt <= 8) return launch_fps_pruned2<8>(xyz, sb, sn, sd, B, N, S, idx, st);
        return launch_fps_pruned2<16>(xyz, sb, sn, sd, B, N, S, idx, st);
    }
    if (ppt <= 2) return launch_fps_plain<2>(xyz, sb, sn, sd, B, N, S, idx, st);
    if (ppt <= 4) return launch_fps_plain<4>(xyz, sb, sn, sd, B, N, S, idx, st);
    if (ppt <= 8) return launch_fps_plain<8>(xyz, sb, sn, sd, B, N, S, idx, st);
    if (ppt <= 16) return launch_fps_plain<16>(xyz, sb, sn, sd, B, N, S, idx, st);
    if (ppt <= 32) return launch_fps_plain<32>(xyz, sb, sn, sd, B, N, S, idx, st);
    return RPE_EUNSUPPORTED;
}

RPE_API int rpe_fps(const float *xyz, int64_t sb, int64_t sn, int64_t sd, int B, int N, int S, int64_t *idx, rpe_stream_t stream) {
    return rpe_fps_algo(xyz, sb, sn, sd, B, N, S, idx, RPE_FPS_AUTO, stream);
}
