// IDS (inverse-depth-scaling) transforms of the clouds for gfx950: perspect2parallel in front of FPS / KNN and
// parallel2perspect behind the 3-D flow head (models/utils.py:320-377, called from models/RPEFlow.py:68-69, 88-93).
//
// The forward transform feeds furthest-point sampling, which is chaotic in its inputs (one different argmax changes
// every later sample), so it is written with the reference's CPU rounding, operation for operation:
//   x' = fl(fl(fl(cx + fl(fl(f / z) * x)) * sw) - hw)          (utils.py:328, 340)
//   z' = fl(fl(fl(f * log z) + 1) * sz)                         (utils.py:330, 342)
// with every scalar (sw = (Wp-1)/(W-1), hw = (Wp-1)/2, ...) rounded to fp32 by the caller, as torch rounds a Python
// float that meets an fp32 tensor.  The one transcendental, log, is the CORRECTLY ROUNDED fp32 logarithm (fp64 log,
// rounded once): the reference's CPU log is MKL's high-accuracy vsLn, which is correctly rounded on all but ~2e-4 of
// its inputs; a device logf (1 ulp) would disagree on several percent of the points.
// The library is built with -ffp-contract=off; divisions are IEEE (hipcc's default for fp32).
#include "common.h"

namespace {

struct IdsScales {
    float sw, sh, hw, hh, sz;  // (Wp-1)/(W-1), (Hp-1)/(H-1), (Wp-1)/2, (Hp-1)/2, min(sw, sh)
};

__device__ __forceinline__ float log_cr(float z) { return (float)log((double)z); }
__device__ __forceinline__ float exp_cr(float v) { return (float)exp((double)v); }

// One thread per point of one cloud; clouds c = 0..n_clouds-1 are channel triples of pcs [B, 3*n_clouds, N];
// out [n_clouds*B, 3, N] contiguous, cloud-major (torch.cat([pc1, pc2], dim=0)).
__global__ __launch_bounds__(256) void ids_forward_kernel(const float *__restrict__ pcs, int64_t sb, int64_t sc, int64_t sn,
                                                          const float *__restrict__ intr, int64_t i_sb, int B, int N,
                                                          IdsScales s, float *__restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y, c = blockIdx.z;
    if (n >= N) return;
    const float f = intr[b * i_sb], cx = intr[b * i_sb + 1], cy = intr[b * i_sb + 2];
    const float *p = pcs + b * sb + (int64_t)(3 * c) * sc + (int64_t)n * sn;
    const float x = p[0], y = p[sc], z = p[2 * sc];
    const float fz = f / z;
    float dx = fz * x;
    dx = cx + dx;
    dx = dx * s.sw;
    dx = dx - s.hw;
    float dy = fz * y;
    dy = cy + dy;
    dy = dy * s.sh;
    dy = dy - s.hh;
    float dz = f * log_cr(z);
    dz = dz + 1.0f;
    dz = dz * s.sz;
    float *o = out + ((int64_t)(c * B + b) * 3) * N + n;
    o[0] = dx;
    o[N] = dy;
    o[2 * (int64_t)N] = dz;
}

// utils.py:349-377 on one point
__device__ __forceinline__ void to_perspective(float x, float y, float z, float f, float cx, float cy, const IdsScales &s,
                                               float &ox, float &oy, float &oz) {
    x = (x + s.hw) / s.sw;
    y = (y + s.hh) / s.sh;
    z = z / s.sz;
    oz = exp_cr((z - 1.0f) / f);
    ox = (x - cx) * oz / f;
    oy = (y - cy) * oz / f;
}

// out = parallel2perspect(xyz + flow) - parallel2perspect(xyz)      (RPEFlow.py:91-93); all [B,3,N]
__global__ __launch_bounds__(256) void ids_flow_inverse_kernel(const float *__restrict__ xyz, int64_t x_sb, int64_t x_sc, int64_t x_sn,
                                                               const float *__restrict__ flow, int64_t f_sb, int64_t f_sc, int64_t f_sn,
                                                               const float *__restrict__ intr, int64_t i_sb, int N, IdsScales s,
                                                               float *__restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= N) return;
    const float f = intr[b * i_sb], cx = intr[b * i_sb + 1], cy = intr[b * i_sb + 2];
    const float *p = xyz + b * x_sb + (int64_t)n * x_sn;
    const float *q = flow + b * f_sb + (int64_t)n * f_sn;
    const float x = p[0], y = p[x_sc], z = p[2 * x_sc];
    const float wx = x + q[0], wy = y + q[f_sc], wz = z + q[2 * f_sc];
    float ax, ay, az, bx, by, bz;
    to_perspective(wx, wy, wz, f, cx, cy, s, ax, ay, az);
    to_perspective(x, y, z, f, cx, cy, s, bx, by, bz);
    float *o = out + ((int64_t)b * 3) * N + n;
    o[0] = ax - bx;
    o[N] = ay - by;
    o[2 * (int64_t)N] = az - bz;
}

}  // namespace

RPE_API int rpe_ids_forward(const float *pcs, int64_t sb, int64_t sc, int64_t sn, const float *intrinsics, int64_t i_sb, int B,
                            int n_clouds, int N, float sw, float sh, float hw, float hh, float sz, float *out,
                            rpe_stream_t stream) {
    if (!pcs || !intrinsics || !out) return RPE_EINVAL;
    if (B < 0 || N < 0 || n_clouds < 1) return RPE_EINVAL;
    if (B == 0 || N == 0) return 0;
    if (B > 65535 || n_clouds > 65535) return RPE_EUNSUPPORTED;
    IdsScales s{sw, sh, hw, hh, sz};
    hipLaunchKernelGGL(ids_forward_kernel, dim3((N + 255) / 256, B, n_clouds), dim3(256), 0, (hipStream_t)stream, pcs, sb, sc, sn,
                       intrinsics, i_sb, B, N, s, out);
    return rpe_launch_status();
}

RPE_API int rpe_ids_flow_inverse(const float *xyz, int64_t x_sb, int64_t x_sc, int64_t x_sn, const float *flow, int64_t f_sb,
                                 int64_t f_sc, int64_t f_sn, const float *intrinsics, int64_t i_sb, int B, int N, float sw, float sh,
                                 float hw, float hh, float sz, float *out, rpe_stream_t stream) {
    if (!xyz || !flow || !intrinsics || !out) return RPE_EINVAL;
    if (B < 0 || N < 0) return RPE_EINVAL;
    if (B == 0 || N == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    IdsScales s{sw, sh, hw, hh, sz};
    hipLaunchKernelGGL(ids_flow_inverse_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, xyz, x_sb, x_sc, x_sn,
                       flow, f_sb, f_sc, f_sn, intrinsics, i_sb, N, s, out);
    return rpe_launch_status();
}
