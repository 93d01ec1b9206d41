// correlation2d forward (81-plane local cost volume) for gfx950.
//
// Replaces correlation_forward_kernel (correlation_forward_kernel.cu:11-54: one
// 32-thread block per pixel, NHWC inputs made by two permute passes in
// wrapper.py:68-69) with kernels that read NCHW directly:
//
//   corr_direct_kernel  any max_displacement; one thread per output element.
//                       Slow, simple; the cross-check for the kernel below.
//   corr_mfma_kernel    max_displacement == 4.  The contraction over channels runs
//                       on the matrix cores with v_mfma_f32_4x4x1_16b_f32: one
//                       instruction = 16 independent 4x4 outer products.  Lane l of
//                       the A operand holds in1[c][y][x0+l], lane l of the B operand
//                       holds in2[c][y+dy][x0+4s+l] for s in {-1,0,+1}: plain
//                       contiguous row segments, no shuffles.  Block g (lanes 4g..4g+3)
//                       then accumulates in1 pixels 4g..4g+3 against in2 pixels
//                       4(g+s)..4(g+s)+3, i.e. dx = 4s + j - i in [-7,7]; the 9 of
//                       12 columns with |dx| <= 4 are kept (75 % of the MFMA work is
//                       useful, against 28 % for a banded 16x16x4 product).
//                       Tiles of in1 and in2 (with a 4-pixel halo) are staged through
//                       LDS per chunk of CK channels; the next chunk's global loads
//                       are issued before the current chunk's MFMAs and written to
//                       LDS after them.
//
// out[b][(dy+md)*(2md+1)+(dx+md)][y][x] = (1/C) sum_c in1[b][c][y][x] * in2[b][c][y+dy][x+dx],
// zero outside the image (wrapper.py:56-65).
#include <type_traits>
#include <utility>

#include "common.h"

#ifndef RPE_CORR_PROBE
#define RPE_CORR_PROBE 0  // 1, 2: diagnostic builds that drop part of corr_mfma_dma_kernel's work (tools/corr_energy_probes.sh), never shipped
#endif
// (not exported: rpe_abi_version() reports it, so that a library whose correlation kernel computes wrong results on purpose
// cannot be loaded by accident through a leaked RPE_HIP_LIB; the ring-depth probes 3-5 compute right results and report 0)
int rpe_diagnostic_flavour() { return (RPE_CORR_PROBE == 1 || RPE_CORR_PROBE == 2) ? RPE_CORR_PROBE : 0; }

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void corr_direct_kernel(const float *__restrict__ in1, const float *__restrict__ in2,
                                                          int C, int H, int W, int md, float slope,
                                                          float *__restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int n = 2 * md + 1;
    const int b = blockIdx.z / (n * n), d = blockIdx.z % (n * n);
    if (x >= W) return;
    const int y2 = y + d / n - md, x2 = x + d % n - md;
    const int64_t HW = (int64_t)H * W;
    float s = 0.f;
    if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) {
        const float *a = in1 + (int64_t)b * C * HW + (int64_t)y * W + x;
        const float *c2 = in2 + (int64_t)b * C * HW + (int64_t)y2 * W + x2;
        // eight independent partial sums: as one chain of C dependent fma + load pairs the small levels (where this
        // kernel runs) took ~58 us for a few MFLOP
        float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int c = 0;
        for (; c + 8 <= C; c += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = __fmaf_rn(a[(c + u) * HW], c2[(c + u) * HW], p[u]);
        }
        for (; c < C; ++c) p[0] = __fmaf_rn(a[c * HW], c2[c * HW], p[0]);
        s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    }
    s = s / (float)C;
    if (slope != 0.f) s = s >= 0.f ? s : s * slope;
    out[((int64_t)b * n * n + d) * HW + (int64_t)y * W + x] = s;
}

// ---- small maps (the three coarse pyramid levels: 9 x 15 ... 36 x 60 pixels, 96 ... 192 channels) --------------------------
// A few hundred pixels: one thread per output element is a handful of half-empty waves, each walking all C channels as a chain
// of memory round trips (18 x 30, C = 128, B = 4: 23 us for 22 MFLOP, on the decoder's critical path).  Here a workgroup takes
// 64 consecutive pixels (rows flattened: no lanes lost to a 15- or 30-pixel row) and ONE displacement row dy; its CG waves
// split the channels, every lane keeps the 2 md + 1 sums of its dy row -- one in1 value against 2 md + 1 neighbouring in2 values
// per channel, several channels' loads in flight -- and the waves' partial sums meet in LDS.  md <= 4.
constexpr int kSmallND = 9;

template <int CG>
__global__ __launch_bounds__(CG * RPE_WAVE) void corr_small_kernel(const float *__restrict__ in1, const float *__restrict__ in2, int C, int H,
                                                                   int W, int md, float slope, float *__restrict__ out) {
    __shared__ float red[CG][kSmallND][RPE_WAVE];
    const int lane = rpe_lane(), grp = threadIdx.x >> 6;
    const int n = 2 * md + 1;
    const int HW = H * W;
    const int p = blockIdx.x * RPE_WAVE + lane, dyi = blockIdx.y, b = blockIdx.z;
    const int y = p / W, x = p - y * W;
    const int y2 = y + dyi - md;
    const bool row_ok = p < HW && y2 >= 0 && y2 < H;
    unsigned ok = 0;  // bit j: the pixel (y2, x + j - md) lies inside the image
#pragma unroll
    for (int j = 0; j < kSmallND; ++j) ok |= (row_ok && j < n && x + j - md >= 0 && x + j - md < W) ? (1u << j) : 0u;
    const int cpg = (C + CG - 1) / CG;
    const int c0 = grp * cpg, c1 = min(C, c0 + cpg);
    const float *a = in1 + (int64_t)b * C * HW + p;
    const float *v = in2 + (int64_t)b * C * HW + (int64_t)y2 * W + (x - md);
    float acc[kSmallND];
#pragma unroll
    for (int j = 0; j < kSmallND; ++j) acc[j] = 0.f;
    constexpr int CU = 4;  // channels per trip
    for (int c = c0; c < c1; c += CU) {
        float av[CU], vv[CU][kSmallND];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const bool cok = c + u < c1;
            av[u] = (cok && ok) ? a[(int64_t)(c + u) * HW] : 0.f;
#pragma unroll
            for (int j = 0; j < kSmallND; ++j) vv[u][j] = (cok && ((ok >> j) & 1u)) ? v[(int64_t)(c + u) * HW + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < CU; ++u)
#pragma unroll
            for (int j = 0; j < kSmallND; ++j) acc[j] = __fmaf_rn(av[u], vv[u][j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < kSmallND; ++j) red[grp][j][lane] = acc[j];
    __syncthreads();
    for (int t = threadIdx.x; t < n * RPE_WAVE; t += CG * RPE_WAVE) {
        const int j = t >> 6, l = t & (RPE_WAVE - 1);
        const int q = blockIdx.x * RPE_WAVE + l;
        if (q >= HW) continue;
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < CG; ++g) sum += red[g][j][l];
        sum = sum / (float)C;
        if (slope != 0.f) sum = sum >= 0.f ? sum : sum * slope;
        out[((int64_t)b * n * n + dyi * n + j) * HW + q] = sum;
    }
}

template <int CG>
void launch_corr_small(const float *in1, const float *in2, int B, int C, int H, int W, int md, float slope, float *out, hipStream_t st) {
    const int n = 2 * md + 1;
    dim3 grid((H * W + RPE_WAVE - 1) / RPE_WAVE, n, B), block(CG * RPE_WAVE);
    hipLaunchKernelGGL(corr_small_kernel<CG>, grid, block, 0, st, in1, in2, C, H, W, md, slope, out);
}

// ---- MFMA kernel ---------------------------------------------------------------
constexpr int MD = 4;       // max displacement this kernel is built for
constexpr int ND = 2 * MD + 1;
constexpr int TX = 64;      // pixels per tile row = lanes
constexpr int PX2 = TX + 2 * MD;  // in2 row segment incl. halo

template <int RY, int NW, int CK>
__global__ __launch_bounds__(NW * RPE_WAVE) void corr_mfma_kernel(const float *__restrict__ in1,
                                                                  const float *__restrict__ in2, int C, int H, int W,
                                                                  float slope, float *__restrict__ out) {
    constexpr int TY = RY * NW;          // in1 rows per workgroup
    constexpr int TY2 = TY + 2 * MD;     // in2 rows incl. halo
    constexpr int NT = NW * RPE_WAVE;
    constexpr int N1 = CK * TY * TX;     // staged in1 floats per chunk
    constexpr int N2 = CK * TY2 * PX2;   // staged in2 floats per chunk
    constexpr int IT1 = N1 / NT;         // exact: CK*RY
    constexpr int IT2 = (N2 + NT - 1) / NT;
    static_assert(N1 % NT == 0, "in1 staging must divide evenly");

    __shared__ float lds[N1 + N2];
    float *l1 = lds, *l2 = lds + N1;

    const int tid = threadIdx.x, lane = rpe_lane();
    const int wave = rpe_uniform(tid >> 6);
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, b = blockIdx.z;
    const int64_t HW = (int64_t)H * W;
    const float *g1 = in1 + (int64_t)b * C * HW;
    const float *g2 = in2 + (int64_t)b * C * HW;

    // per-thread staging slots: element offset inside one channel plane, or -1
    int o1[IT1], o2[IT2];
    int c1[IT1], c2[IT2];  // channel inside the chunk
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int e = it * NT + tid;
        const int px = e % TX, row = (e / TX) % TY;
        c1[it] = e / (TX * TY);
        const int y = y0 + row, x = x0 + px;
        o1[it] = (y < H && x < W) ? y * W + x : -1;
    }
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
        const int e = it * NT + tid;
        const int col = e % PX2, row = (e / PX2) % TY2;
        c2[it] = e / (PX2 * TY2);
        const int y = y0 - MD + row, x = x0 - MD + col;
        o2[it] = (e < N2 && y >= 0 && y < H && x >= 0 && x < W) ? y * W + x : -1;
    }

    float r1[IT1], r2[IT2];
    auto fetch = [&](int cbase) {
#pragma unroll
        for (int it = 0; it < IT1; ++it) {
            const int c = cbase + c1[it];
            r1[it] = (o1[it] >= 0 && c < C) ? g1[(int64_t)c * HW + o1[it]] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
            const int c = cbase + c2[it];
            r2[it] = (o2[it] >= 0 && c < C) ? g2[(int64_t)c * HW + o2[it]] : 0.f;
        }
    };

    f32x4 acc[RY][ND][3];
#pragma unroll
    for (int ry = 0; ry < RY; ++ry)
#pragma unroll
        for (int dy = 0; dy < ND; ++dy)
#pragma unroll
            for (int s = 0; s < 3; ++s) acc[ry][dy][s] = (f32x4){0.f, 0.f, 0.f, 0.f};

    fetch(0);
    for (int cbase = 0; cbase < C; cbase += CK) {
        __syncthreads();  // everyone finished reading the previous chunk
#pragma unroll
        for (int it = 0; it < IT1; ++it) l1[it * NT + tid] = r1[it];
#pragma unroll
        for (int it = 0; it < IT2; ++it)
            if (it * NT + tid < N2) l2[it * NT + tid] = r2[it];
        __syncthreads();
        if (cbase + CK < C) fetch(cbase + CK);  // in flight while the MFMAs run

#pragma unroll
        for (int c = 0; c < CK; ++c) {
            float a[RY];
#pragma unroll
            for (int ry = 0; ry < RY; ++ry) a[ry] = l1[(c * TY + wave * RY + ry) * TX + lane];
#pragma unroll
            for (int r = 0; r < RY + 2 * MD; ++r) {
                const float *row = l2 + (c * TY2 + wave * RY + r) * PX2 + lane;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const float bv = row[4 * s];
#pragma unroll
                    for (int ry = 0; ry < RY; ++ry) {
                        const int dy = r - ry;  // in2 row (y + dy - MD) sits at local row ry + dy
                        if (dy >= 0 && dy < ND)
                            acc[ry][dy][s] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[ry], bv, acc[ry][dy][s], 0, 0, 0);
                    }
                }
            }
        }
    }

    // D layout: lane 4g+j, register i  <->  in1 pixel x0+4g+i against in2 pixel x0+4(g+s-1)+j
    const int g = lane >> 2, j = lane & 3;
    const float fc = (float)C;
#pragma unroll
    for (int ry = 0; ry < RY; ++ry) {
        const int y = y0 + wave * RY + ry;
        if (y >= H) continue;
#pragma unroll
        for (int dy = 0; dy < ND; ++dy)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int dx = 4 * (s - 1) + j - i;
                    const int x = x0 + 4 * g + i;
                    if (dx >= -MD && dx <= MD && x < W) {
                        float v = acc[ry][dy][s][i] / fc;
                        if (slope != 0.f) v = v >= 0.f ? v : v * slope;
                        out[((int64_t)b * ND * ND + dy * ND + (dx + MD)) * HW + (int64_t)y * W + x] = v;
                    }
                }
    }
}


// ---- MFMA kernel, v2: LDS-DMA ring ------------------------------------------------------
// Same MFMA formulation as corr_mfma_kernel; what changes is how operands reach LDS:
//   * tiles of in1 (TY rows x 64 px) and in2 (TY+8 rows x 72 px, 4-px halo) for a chunk of CK
//     channels travel global -> LDS by global_load_lds_dwordx4 (1 KiB per wave-instruction, no
//     staging registers).  Out-of-image pieces are not masked: their lanes read a 16-byte zero
//     word, so every wave issues exactly PPW DMAs per chunk and the counted s_waitcnt vmcnt(N)
//     below is exact;
//   * a ring of NSLOT LDS slots: chunks k+1..k+NSLOT-1 are in flight while chunk k feeds the
//     MFMAs; one raw s_barrier per chunk (after it everybody has finished chunk k-1, whose slot
//     the next DMAs overwrite);
//   * PD > 0: the operand reads of one (channel, in2 row) step are issued PD steps ahead of the
//     MFMAs that consume them (a register ring of PD+1 x 3 B operands, A operands double-buffered),
//     across channel boundaries inside a chunk; sched_group_barrier pins "3 LDS reads, then the
//     step's MFMAs"; the next chunk's DMAs are issued one piece every few steps instead of as a
//     burst after the barrier (a DMA issue costs the wave tens of cycles, which its SIMD partner
//     fills with MFMAs as long as the two do not burst together);
//   * epilogue through LDS: each wave transposes its accumulators to [dx][64 px] rows and stores
//     256-byte coalesced segments (v1 stores 4-byte pieces at a 16-byte stride).
// With RY = 2 the 216 accumulators leave room for two waves per SIMD (8 waves x 2 rows = 16 rows
// per workgroup); RY = 4 (432 accumulators) does not survive hipcc's AGPR/VGPR split.
// Requires W % 4 == 0, C % CK == 0 and 16-byte aligned inputs (whole float4 pieces in or out of
// the image); anything else takes corr_mfma_kernel.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ float4 g_corr_zero16;  // zero-initialised; the source of every out-of-image DMA lane

// f(integral_constant<int, 0>{}), ..., f(integral_constant<int, N-1>{}): a loop whose index is a
// constant expression (sched_group_barrier needs immediates)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int RY, int NW, int CK, int NSLOT>
struct V2 {
    static constexpr int TY = RY * NW;              // in1 rows per workgroup
    static constexpr int TY2 = TY + 2 * MD;         // in2 rows (halo)
    static constexpr int F1 = CK * TY * (TX / 4);   // float4 elements of the in1 tile
    static constexpr int F2 = CK * TY2 * (PX2 / 4); // ... of the in2 tile
    static constexpr int PIECES1 = (F1 + 63) / 64;  // a DMA piece = 64 lanes x 16 B
    static constexpr int PIECES2 = (F2 + 63) / 64;
    static constexpr int PPW = (PIECES1 + PIECES2 + NW - 1) / NW;  // pieces per wave per chunk (exact, incl. padding)
    static constexpr int OFF2 = PIECES1 * 256;      // float offset of the in2 tile inside a slot
    static constexpr int SLOT = PPW * NW * 256;     // floats per ring slot
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)NSLOT * SLOT;
    static_assert(F1 % 64 == 0, "in1 tile must be whole pieces so that the in2 tile starts piece-aligned");
    static_assert(NSLOT * SLOT >= NW * 2 * ND * TX, "epilogue staging must fit in the ring");
};

template <int RY, int NW, int CK, int NSLOT, int PD, bool SPREAD>
__global__ __launch_bounds__(NW * RPE_WAVE) void corr_mfma_dma_kernel(const float *__restrict__ in1,
                                                                      const float *__restrict__ in2, int nbatch, int C, int H,
                                                                      int W, float slope, float *__restrict__ out) {
    using G = V2<RY, NW, CK, NSLOT>;
    extern __shared__ float lds[];  // the ONLY LDS object of this kernel: [NSLOT][SLOT]
    const int tid = threadIdx.x, lane = rpe_lane();
    const int wave = rpe_uniform(tid >> 6);
    // Tile order.  Workgroups are dealt round-robin over the 8 XCDs (ids i and i+8 share one, each with
    // its own L2), so XCD k is given the contiguous range [k*per, (k+1)*per) of the ROW-major tile
    // order: the ~32 tiles an XCD runs at once are then about two full tile rows.  That matters
    // because an in2 segment [x0-4, x0+68) spans four 128-byte lines of which two are also read by
    // the left and two by the right neighbour (and 8 of 24 rows by the tile below): un-shared, in2 is
    // fetched 3x (measured: fabric traffic 1.75 GB -> 1.20 GB on the 1x256x544x960 case).
    // Speed only; any placement gives the same result.
    const int nty = (H + G::TY - 1) / G::TY, ntx = (W + TX - 1) / TX;
    const int per = gridDim.x / 8;
    const int tile = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if (tile >= ntx * nty * nbatch) return;  // padding workgroups of the remap (uniform exit, before any barrier)
    const int b = tile / (ntx * nty), tr = tile % (ntx * nty);
    const int x0 = (tr % ntx) * TX, y0 = (tr / ntx) * G::TY;
    const int64_t HW = (int64_t)H * W;
    const float *g1 = in1 + (int64_t)b * C * HW;
    const float *g2 = in2 + (int64_t)b * C * HW;
    const float *zero = reinterpret_cast<const float *>(&g_corr_zero16);

    // this lane's source offset (floats, inside a CK-channel chunk) for each of its wave's pieces
    int off[G::PPW];  // < 0: outside the image or padding -> zero word  (CK*H*W < 2^31 checked by the launcher)
#pragma unroll
    for (int t = 0; t < G::PPW; ++t) {
        const int q = wave + NW * t;  // piece index inside the slot, wave-uniform
        int o = -1;
        if (q < G::PIECES1) {
            const int f = q * 64 + lane;  // float4 index in [CK][TY][16]
            const int c = f / (G::TY * 16), r = (f / 16) % G::TY, c4 = f % 16;
            const int y = y0 + r, x = x0 + 4 * c4;
            if (y < H && x < W) o = c * (H * W) + y * W + x;
        } else {
            const int f = (q - G::PIECES1) * 64 + lane;  // float4 index in [CK][TY2][18]
            const int c = f / (G::TY2 * 18), r = (f / 18) % G::TY2, c4 = f % 18;
            const int y = y0 - MD + r, x = x0 - MD + 4 * c4;
            if (f < G::F2 && y >= 0 && y < H && x >= 0 && x < W) o = c * (H * W) + y * W + x;
        }
        off[t] = o;
    }

    auto issue_piece = [&](int chunk, int slot_index, int t) {  // t is a compile-time constant at every call site
        float *slot = lds + slot_index * G::SLOT;
        const int64_t cbase = (int64_t)chunk * CK * HW;
        const int q = wave + NW * t;
        // both addresses are formed unconditionally and selected: no branch in the chunk body
        const float *inside = (q >= G::PIECES1 ? g2 : g1) + cbase + max(off[t], 0);
        const float *src = off[t] >= 0 ? inside : zero;
        __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(slot + q * 256), 16, 0, 0);
    };
    auto issue = [&](int chunk, int slot_index) {
#pragma unroll
        for (int t = 0; t < G::PPW; ++t) issue_piece(chunk, slot_index, t);
    };

    f32x4 acc[RY][ND][3];
#pragma unroll
    for (int ry = 0; ry < RY; ++ry)
#pragma unroll
        for (int dy = 0; dy < ND; ++dy)
#pragma unroll
            for (int s = 0; s < 3; ++s) acc[ry][dy][s] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = C / CK;
    constexpr int AHEAD = NSLOT - 1;  // chunks in flight beyond the one being consumed
    static_assert(AHEAD >= 1 && AHEAD <= 6, "ring depth supported by the wait table below");
    static_assert((AHEAD - 1) * G::PPW <= 63, "vmcnt is a 6-bit counter");
#pragma unroll
    for (int p = 0; p < AHEAD; ++p) issue(min(p, nchunks - 1), p);
    for (int ch = 0; ch < nchunks; ++ch) {
        // Chunk ch has landed once at most the DMAs of the AHEAD-1 younger chunks of THIS wave are pending.  Every
        // iteration issues exactly PPW DMAs -- past the end they re-fetch the last chunk into a slot nobody reads
        // any more -- so the count is a constant and the chunk body has no branch (which lets hipcc count its
        // lgkmcnt waits instead of draining the LDS queue at every basic-block boundary).
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * G::PPW) : "memory");
        __builtin_amdgcn_s_barrier();  // ... for every wave; and everyone is done reading chunk ch-1
        const int next = min(ch + AHEAD, nchunks - 1);
        const int next_slot = (ch + AHEAD) % NSLOT;
        constexpr bool more = true;
        if (!(PD > 0 && SPREAD) && more) issue(next, next_slot);  // SPREAD: issued piece by piece among the MFMA steps below

        const int slot_id = ch % NSLOT;
        const float *l1 = lds + slot_id * G::SLOT + (wave * RY) * TX + lane;
        const float *l2 = lds + slot_id * G::SLOT + G::OFF2 + (wave * RY) * PX2 + lane;
        static_assert(PD > 0, "the operand reads run PD steps ahead of the MFMAs");
        {
            constexpr int ROWS = RY + 2 * MD, STEPS = CK * ROWS;
            constexpr int DMA_EVERY = STEPS / G::PPW;
            static_assert(DMA_EVERY >= 1, "more DMA pieces than steps");
            float br[PD + 1][3], ar[2][RY];
            {
                auto load_step = [&](int t) {  // compile-time t after inlining
                    const int c = t / ROWS, r = t % ROWS;
                    if (r == 0) {
#pragma unroll
                        for (int ry = 0; ry < RY; ++ry) ar[c & 1][ry] = l1[(c * G::TY + ry) * TX];
                    }
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
#if RPE_CORR_PROBE == 2  // energy probe (WRONG results): one B-operand read per step instead of three, the matrix work unchanged
                        if (s > 0) {
                            br[t % (PD + 1)][s] = br[t % (PD + 1)][0];
                            continue;
                        }
#endif
                        br[t % (PD + 1)][s] = l2[(c * G::TY2 + r) * PX2 + 4 * s];
                    }
                };
#pragma unroll
                for (int t = 0; t < PD; ++t) load_step(t);
                static_for<STEPS>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr int c = t / ROWS, r = t % ROWS;
                    if constexpr (SPREAD && t % DMA_EVERY == DMA_EVERY / 2 && t / DMA_EVERY < G::PPW) {
                        if (more) issue_piece(next, next_slot, t / DMA_EVERY);
                    }
                    if constexpr (t + PD < STEPS) load_step(t + PD);
#pragma unroll
                    for (int s = 0; s < 3; ++s)
#pragma unroll
                        for (int ry = 0; ry < RY; ++ry) {
                            const int dy = r - ry;
#if RPE_CORR_PROBE == 1  // energy probe (tools/corr_energy_probes.sh; WRONG results): a third of the matrix work, every operand read kept
                            if (s != 1) {
                                asm volatile("" ::"v"(br[t % (PD + 1)][s]));
                                continue;
                            }
#endif
                            if (dy >= 0 && dy < ND)
                                acc[ry][dy][s] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[c & 1][ry], br[t % (PD + 1)][s], acc[ry][dy][s], 0, 0, 0);
                        }
                    constexpr int n_valid = (r < RY ? r + 1 : (r >= ND ? ROWS - r : RY));  // in1 rows this in2 row pairs with
                    if constexpr (t + PD < STEPS)
                        __builtin_amdgcn_sched_group_barrier(0x100, ((t + PD) % ROWS == 0) ? 3 + RY : 3, 0);  // DS reads
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * n_valid, 0);                             // MFMAs
                });
            }
        }
    }

    // ---- epilogue: per wave and (row, dy), a [9 dx][64 px] staging tile in LDS, then coalesced stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (unused) DMAs have landed
    __builtin_amdgcn_s_barrier();                     // ... in every wave, and all waves left the ring
    float *stage = lds + wave * (2 * ND * TX);  // two tiles per wave, used alternately
    const int g = lane >> 2, j = lane & 3;
    const float fc = (float)C;
#pragma unroll
    for (int ry = 0; ry < RY; ++ry) {
        const int y = y0 + wave * RY + ry;
#pragma unroll
        for (int dy = 0; dy < ND; ++dy) {
            float *tile_lds = stage + ((ry * ND + dy) & 1) * (ND * TX);
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int dx = 4 * (s - 1) + j - i;
                    if (dx >= -MD && dx <= MD) {
                        float v = acc[ry][dy][s][i] / fc;
                        if (slope != 0.f) v = v >= 0.f ? v : v * slope;
                        tile_lds[(dx + MD) * TX + 4 * g + i] = v;
                    }
                }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's tile is written
            __builtin_amdgcn_wave_barrier();
            if (y < H && x0 + lane < W) {
                float *o = out + ((int64_t)b * ND * ND + dy * ND) * HW + (int64_t)y * W + x0 + lane;
#pragma unroll
                for (int d = 0; d < ND; ++d) o[(int64_t)d * HW] = tile_lds[d * TX + lane];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int RY, int NW, int CK, int NSLOT, int PD, bool SPREAD>
int launch_mfma_dma(const float *in1, const float *in2, int B, int C, int H, int W, float slope, float *out, hipStream_t st) {
    using G = V2<RY, NW, CK, NSLOT>;
    if (C % CK != 0 || (int64_t)H * W * CK >= (1ll << 31)) return RPE_EUNSUPPORTED;
    auto kern = corr_mfma_dma_kernel<RY, NW, CK, NSLOT, PD, SPREAD>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    const int64_t tiles = (int64_t)((W + TX - 1) / TX) * ((H + G::TY - 1) / G::TY) * B;
    if (tiles > (1 << 28)) return RPE_EUNSUPPORTED;
    dim3 grid((unsigned)((tiles + 7) / 8 * 8)), block(NW * RPE_WAVE);  // multiple of 8: see the tile-order note
    hipLaunchKernelGGL(kern, grid, block, G::LDS_BYTES, st, in1, in2, B, C, H, W, slope, out);
    return 0;
}


#ifdef RPE_EXPERIMENTAL
__global__ void probe_mfma4x4_kernel(float *out) {
    const int lane = threadIdx.x;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)lane, 100.f * (float)lane, d, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = d[i];
}

#endif

template <int RY, int NW, int CK>
void launch_mfma(const float *in1, const float *in2, int B, int C, int H, int W, float slope, float *out, hipStream_t st) {
    constexpr int TY = RY * NW;
    dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY, B), block(NW * RPE_WAVE);
    hipLaunchKernelGGL((corr_mfma_kernel<RY, NW, CK>), grid, block, 0, st, in1, in2, C, H, W, slope, out);
}

}  // namespace

RPE_API int rpe_correlation2d_forward(const float *in1, const float *in2, int B, int C, int H, int W, int md,
                                      float leaky_slope, int algo, float *out, rpe_stream_t stream) {
    if (!in1 || !in2 || !out || B < 0 || C <= 0 || H <= 0 || W <= 0 || md < 0) return RPE_EINVAL;
    if ((int64_t)H * W >= (1ll << 31)) return RPE_EUNSUPPORTED;
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int n = 2 * md + 1;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(in2)) & 15) == 0;
    // pick (measured on MI355X, tools/corr_gate_table.py -> profiles/r06_corr_gate_table*.txt): the LDS-DMA ring kernels where their
    // alignment conditions hold, from 72x120 maps up -- TWO rows a wave (216 accumulators, two waves a SIMD: the operand reuse
    // that carries the big maps) once that tiling puts a wave on every SIMD, ONE row a wave (108 accumulators, twice the waves)
    // below, where the launch is latency-bound: 4 x 64 x 72 x 120 51 -> 33 us, 3 x 32 x 128 x 160 36 -> 24 us, and 39 against 44 us
    // the other way at 4 x 32 x 144 x 240 (1152 waves) -- the register-staged MFMA kernel for other large maps, the small-map
    // kernel up to 8 x 36 x 60 pixels (36x60, C = 96: 15 us against 42-72 for the tiled kernels), one thread per output for the rest.
    if (algo == 0) {
        const bool dma_ok = md == MD && W % 4 == 0 && aligned && B <= 65535 && (int64_t)H * W * 4 < (1ll << 31);
        const int64_t px = (int64_t)H * W;
        const int64_t waves2 = (int64_t)B * ((H + 1) / 2) * ((W + TX - 1) / TX);  // waves of the two-rows-a-wave tiling; the chip has 1024 SIMDs
        if (dma_ok && C % 4 == 0 && px >= 72 * 120 && waves2 < 1024) algo = 8;
        else if (dma_ok && C % 2 == 0 && px >= 72 * 120) algo = 7;
        else if (md == MD && px >= 72 * 120) algo = 2;
        else if (md <= 4 && B <= 65535 && (int64_t)B * px <= 8 * 36 * 60) algo = 3;  // the maps it was written and measured on (<= 36x60, 2B = 8)
        else algo = 1;
    }
    if (algo == 2) {
        if (md != MD) return RPE_EUNSUPPORTED;
        if (B > 65535) return RPE_EUNSUPPORTED;
        launch_mfma<2, 4, 4>(in1, in2, B, C, H, W, leaky_slope, out, st);
    } else if (algo == 7 || algo == 8) {
        if (md != MD || B > 65535 || W % 4 != 0 || !aligned) return RPE_EUNSUPPORTED;
        int rc = algo == 8 ? launch_mfma_dma<1, 8, 4, 3, 3, true>(in1, in2, B, C, H, W, leaky_slope, out, st)  // one row a wave
#if RPE_CORR_PROBE >= 3 && RPE_CORR_PROBE <= 5  // ring depth probes: 4 / 5 / 6 slots instead of 3 (tools/corr_energy_probes.sh)
                           : launch_mfma_dma<2, 8, 2, RPE_CORR_PROBE + 1, 3, true>(in1, in2, B, C, H, W, leaky_slope, out, st);
#else
                           : launch_mfma_dma<2, 8, 2, 3, 3, true>(in1, in2, B, C, H, W, leaky_slope, out, st);
#endif
        if (rc) return rc;
    } else if (algo == 3) {  // small maps: 64 flattened pixels x one displacement row a workgroup, the channels split over its waves
        if (md > 4 || B > 65535) return RPE_EUNSUPPORTED;
        if (C <= 48) launch_corr_small<4>(in1, in2, B, C, H, W, md, leaky_slope, out, st);
        else if (C <= 128) launch_corr_small<8>(in1, in2, B, C, H, W, md, leaky_slope, out, st);
        else launch_corr_small<16>(in1, in2, B, C, H, W, md, leaky_slope, out, st);
    } else if (algo == 1) {
        if ((int64_t)B * n * n > 65535 || H > 65535) return RPE_EUNSUPPORTED;
        dim3 grid((W + 255) / 256, H, B * n * n), block(256);
        hipLaunchKernelGGL(corr_direct_kernel, grid, block, 0, st, in1, in2, C, H, W, md, leaky_slope, out);
    } else {
        return RPE_EINVAL;
    }
    return rpe_launch_status();
}

#ifdef RPE_EXPERIMENTAL
RPE_API int rpe_probe_mfma4x4(float *out256, rpe_stream_t stream) {
    if (!out256) return RPE_EINVAL;
    hipLaunchKernelGGL(probe_mfma4x4_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out256);
    return rpe_launch_status();
}
#endif
