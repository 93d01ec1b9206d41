// correlation2d forward (81-plane local cost volume) for gfx950.
//
// Replaces correlation_forward_kernel (correlation_forward_kernel.cu:11-54: one
// 32-thread block per pixel, NHWC inputs made by two permute passes in
// wrapper.py:68-69) with kernels that read NCHW directly:
//
//   corr_direct_kernel  any max_displacement; one thread per output element.
//                       Slow, simple; the cross-check for the kernel below.
//   corr_row_kernel     max_displacement 1..4, W <= 256: one workgroup per image row, VALU, LDS-staged;
//                       what every in-model call uses.
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
#include "common.h"

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
        for (int c = 0; c < C; ++c) s = __fmaf_rn(a[c * HW], c2[c * HW], s);
    }
    s = s / (float)C;
    if (slope != 0.f) s = s >= 0.f ? s : s * slope;
    out[((int64_t)b * n * n + d) * HW + (int64_t)y * W + x] = s;
}


// ---- row kernel: small and medium maps (every in-model call) ------------------------
// One workgroup per (batch, image row): the row of in1 and the 2md+1 rows of in2 it needs
// are staged in LDS per chunk of CK channels; thread t owns pixel x = t % XW and the
// displacement rows dy = g, g+G, ... of group g = t / XW, all 2md+1 dx of them in
// registers (accumulators, LDS offsets are immediates).  Every output of the row is
// produced by exactly one thread in channel order: deterministic, no atomics.
// Launch-/LDS-bound by design: the in-model maps are at most 144x240x32 (0.3 GFLOP a
// sample in total), far below anything the MFMA tile kernel can fill the chip with.
template <int MDT, int G>
__global__ __launch_bounds__(256) void corr_row_kernel(const float *__restrict__ in1, const float *__restrict__ in2,
                                                       int C, int H, int W, float slope, float *__restrict__ out) {
    constexpr int N = 2 * MDT + 1;
    constexpr int XW = 256 / G;            // threads (pixels) per displacement group
    constexpr int NDY = (N + G - 1) / G;   // displacement rows per thread
    constexpr int CK = 4;
    extern __shared__ float lds[];
    const int Wp = W + 2 * MDT;
    float *l1 = lds;                 // [CK][W]
    float *l2 = lds + CK * W;        // [CK][N][Wp]
    const int tid = threadIdx.x, y = blockIdx.x, b = blockIdx.y;
    const int x = tid % XW, g = tid / XW;
    const int64_t HW = (int64_t)H * W;
    const float *g1 = in1 + (int64_t)b * C * HW + (int64_t)y * W;
    const float *g2 = in2 + (int64_t)b * C * HW;

    float acc[NDY][N];
#pragma unroll
    for (int i = 0; i < NDY; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) acc[i][j] = 0.f;

    const bool active = x < W;
    const int xr = active ? x : 0;
    for (int c0 = 0; c0 < C; c0 += CK) {
        __syncthreads();
        for (int e = tid; e < CK * W; e += 256) {
            const int c = e / W, px = e - c * W;
            l1[e] = (c0 + c < C) ? g1[(int64_t)(c0 + c) * HW + px] : 0.f;
        }
        for (int e = tid; e < CK * N * Wp; e += 256) {
            const int c = e / (N * Wp), r = e - c * (N * Wp);
            const int dy = r / Wp, px = r - dy * Wp;
            const int y2 = y + dy - MDT, x2 = px - MDT;
            const bool ok = (c0 + c < C) && y2 >= 0 && y2 < H && x2 >= 0 && x2 < W;
            l2[e] = ok ? g2[(int64_t)(c0 + c) * HW + (int64_t)y2 * W + x2] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CK; ++c) {
            const float a = l1[c * W + xr];
#pragma unroll
            for (int i = 0; i < NDY; ++i) {
                const int dy = g + i * G;
                if (dy < N) {
                    const float *row = l2 + (c * N + dy) * Wp + xr;
#pragma unroll
                    for (int j = 0; j < N; ++j) acc[i][j] = __fmaf_rn(a, row[j], acc[i][j]);
                }
            }
        }
    }
    if (!active) return;
    const float fc = (float)C;
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
        const int dy = g + i * G;
        if (dy < N) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float v = acc[i][j] / fc;
                if (slope != 0.f) v = v >= 0.f ? v : v * slope;
                out[((int64_t)b * N * N + dy * N + j) * HW + (int64_t)y * W + x] = v;
            }
        }
    }
}

template <int MDT>
int launch_row(const float *in1, const float *in2, int B, int C, int H, int W, float slope, float *out, hipStream_t st) {
    constexpr int N = 2 * MDT + 1;
    const size_t shmem = sizeof(float) * 4 * ((size_t)W + (size_t)N * (W + 2 * MDT));
    dim3 grid(H, B), block(256);
    auto launch = [&](auto kern) -> int {
        if (shmem > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(kern, grid, block, shmem, st, in1, in2, C, H, W, slope, out);
        return 0;
    };
    if (W <= 64) return launch(corr_row_kernel<MDT, 4>);
    if (W <= 128) return launch(corr_row_kernel<MDT, 2>);
    return launch(corr_row_kernel<MDT, 1>);
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

__global__ void probe_mfma4x4_kernel(float *out) {
    const int lane = threadIdx.x;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)lane, 100.f * (float)lane, d, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = d[i];
}

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
    // pick: row kernel for every in-model size, MFMA tiles for large maps, direct for anything else
    if (algo == 0) {
        const bool row_ok = md >= 1 && md <= 4 && W <= 256 && H <= 65535;
        if (md == MD && (int64_t)H * W > 256 * 256) algo = 2;
        else algo = row_ok ? 3 : ((md == MD) ? 2 : 1);
    }
    if (algo == 3) {
        if (W > 256 || H > 65535 || B > 65535) return RPE_EUNSUPPORTED;
        int rc;
        switch (md) {
            case 1: rc = launch_row<1>(in1, in2, B, C, H, W, leaky_slope, out, st); break;
            case 2: rc = launch_row<2>(in1, in2, B, C, H, W, leaky_slope, out, st); break;
            case 3: rc = launch_row<3>(in1, in2, B, C, H, W, leaky_slope, out, st); break;
            case 4: rc = launch_row<4>(in1, in2, B, C, H, W, leaky_slope, out, st); break;
            default: return RPE_EUNSUPPORTED;
        }
        if (rc) return rc;
    } else if (algo == 2) {
        if (md != MD) return RPE_EUNSUPPORTED;
        if (B > 65535) return RPE_EUNSUPPORTED;
        launch_mfma<2, 4, 4>(in1, in2, B, C, H, W, leaky_slope, out, st);
    } else if (algo == 1) {
        if ((int64_t)B * n * n > 65535 || H > 65535) return RPE_EUNSUPPORTED;
        dim3 grid((W + 255) / 256, H, B * n * n), block(256);
        hipLaunchKernelGGL(corr_direct_kernel, grid, block, 0, st, in1, in2, C, H, W, md, leaky_slope, out);
    } else {
        return RPE_EINVAL;
    }
    return rpe_launch_status();
}

RPE_API int rpe_probe_mfma4x4(float *out256, rpe_stream_t stream) {
    if (!out256) return RPE_EINVAL;
    hipLaunchKernelGGL(probe_mfma4x4_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out256);
    return rpe_launch_status();
}
