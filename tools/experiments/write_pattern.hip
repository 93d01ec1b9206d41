// Store (and load) pattern of the 1x1 kernel without its arithmetic: what does the memory system give a launch that writes
// [B][C][P] floats in the order pointwise_conv_kernel does -- a wave = OT output tiles (16 channels each) x PG groups of 64
// positions, one float4 per lane and (tile, register, group): per store instruction 4 channel planes x 256 contiguous bytes --
// and what would it give longer runs per plane?  tools/experiments/write_pattern.py drives it (hipcc -shared, ctypes).
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int OT, int PG>
__global__ __launch_bounds__(512) void pattern_kernel(const float *__restrict__ x, int Cin, float *__restrict__ y, int Cout, int64_t P, int read_x) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int b = blockIdx.z;
    const int ot0 = (blockIdx.y * nw + wave) * OT;
    const int64_t p0 = (int64_t)blockIdx.x * 64 * PG + 4 * j;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (read_x) {  // the B-operand loads: lane (g, j) reads 4 positions of channel 4 kt + g, for every group
        for (int c = g; c < Cin; c += 4)
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                const int64_t p = p0 + 64 * pg;
                if (p < P) {
                    const float4 v = *reinterpret_cast<const float4 *>(x + ((int64_t)b * Cin + c) * P + p);
                    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
                }
            }
    }
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = 16 * (ot0 + o) + 4 * g + r;
            if (oc >= Cout) continue;
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                const int64_t p = p0 + 64 * pg;
                if (p < P) *reinterpret_cast<float4 *>(y + ((int64_t)b * Cout + oc) * P + p) = acc;
            }
        }
}

template <int OT, int PG>
static int launch(const float *x, int Cin, float *y, int Cout, int64_t P, int B, int nw, int read_x, hipStream_t st) {
    const int otiles = (Cout + 15) / 16;
    dim3 grid((unsigned)((P + 64 * PG - 1) / (64 * PG)), (unsigned)((otiles + nw * OT - 1) / (nw * OT)), (unsigned)B);
    hipLaunchKernelGGL((pattern_kernel<OT, PG>), grid, dim3(64 * nw), 0, st, x, Cin, y, Cout, P, read_x);
    return (int)hipGetLastError();
}

extern "C" int write_pattern(const float *x, int Cin, float *y, int Cout, int64_t P, int B, int OT, int PG, int nw, int read_x, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (OT == 4 && PG == 1) return launch<4, 1>(x, Cin, y, Cout, P, B, nw, read_x, st);
    if (OT == 2 && PG == 2) return launch<2, 2>(x, Cin, y, Cout, P, B, nw, read_x, st);
    if (OT == 1 && PG == 4) return launch<1, 4>(x, Cin, y, Cout, P, B, nw, read_x, st);
    if (OT == 2 && PG == 1) return launch<2, 1>(x, Cin, y, Cout, P, B, nw, read_x, st);
    if (OT == 1 && PG == 1) return launch<1, 1>(x, Cin, y, Cout, P, B, nw, read_x, st);
    if (OT == 4 && PG == 2) return launch<4, 2>(x, Cin, y, Cout, P, B, nw, read_x, st);
    if (OT == 4 && PG == 4) return launch<4, 4>(x, Cin, y, Cout, P, B, nw, read_x, st);
    return -1;
}
