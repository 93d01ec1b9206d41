/*
 * rpeflow_hip.h -- C ABI of librpeflow_hip.so: RPEFlow's hot-path operators as
 * hand-written HIP kernels for gfx950 (MI355X).
 *
 * This is the drop-in boundary.  Every entry point takes plain device
 * pointers, sizes, element strides and the HIP stream to launch on; the
 * library owns no memory and never synchronises.  Outputs are fully written
 * (callers may pass uninitialised buffers).  All floating point is fp32, all
 * indices int64, exactly as the reference's extension returns them.
 *
 * Return value: 0 on success, a positive hipError_t if a HIP call failed,
 * a negative RPE_E* code if the arguments are unsupported.  rpe_error_string()
 * names either.
 *
 * Reference interfaces replaced (paths relative to the RPEFlow checkout):
 *   models/csrc/correlation/correlation.cpp:3-4            correlation_forward_kernel_wrapper
 *   models/csrc/k_nearest_neighbor/k_nearest_neighbor.cpp:3-4  k_nearest_neighbor_{2d,3d}_kernel_wrapper
 *   models/csrc/furthest_point_sampling/furthest_point_sampling.cpp:3  furthest_point_sampling_kernel_wrapper
 * plus fused forms of the pure-PyTorch gather/warp/PointConv arithmetic in
 * models/utils.py, models/pointconv.py and models/pwc3d_core.py (cited per entry).
 *
 * Results follow the reference's CPU/PyTorch fallback (models/csrc/wrapper.py),
 * not its CUDA kernels, wherever the two differ (distance form, tie rules).
 */
#ifndef RPEFLOW_HIP_H
#define RPEFLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPE_ABI_VERSION 1

#define RPE_EINVAL (-1)       /* bad size / null pointer */
#define RPE_EUNSUPPORTED (-2) /* valid request this build has no kernel for */

typedef void *rpe_stream_t; /* a hipStream_t; NULL = the legacy default stream */

int rpe_abi_version(void);
const char *rpe_error_string(int code);

/* ---- k_nearest_neighbor ---------------------------------------------------
 * Replaces k_nearest_neighbor_{2d,3d}_kernel_wrapper(b, n, m, k, query, input, idx)
 * (k_nearest_neighbor.cpp:3-4, kernels k_nearest_neighbor_kernel.cu:8-112) with
 * the arithmetic of the CPU fallback (wrapper.py:40-52,115-117):
 *   d = fl(fl(-2*dot + |q|^2) + |p|^2), dot = fma(q2,p2,fma(q1,p1,q0*p0)),
 * k smallest, ascending, equal distances ordered by input index.
 * Either point layout is accepted through strides (elements, not bytes):
 *   input[b][m][d] = input[b*in_sb + m*in_sn + d*in_sd], same for query,
 * so channel-first callers need no transpose (wrapper.py:119-122 does one).
 * idx  [B,Q,k] int64 contiguous; dist [B,Q,k] fp32 contiguous or NULL.
 * Limits: 1 <= D <= 3, 1 <= k <= 64, k <= M.                                  */
int rpe_knn(const float *input, int64_t in_sb, int64_t in_sn, int64_t in_sd,
            const float *query, int64_t q_sb, int64_t q_sn, int64_t q_sd,
            int B, int M, int Q, int D, int k,
            int64_t *idx, float *dist, rpe_stream_t stream);

/* ---- squared_distance (wrapper.py:40-52) ------------------------------------
 * out[b][i][j] = distance above between xyz1[b][i] and xyz2[b][j]; out contiguous. */
int rpe_squared_distance(const float *xyz1, int64_t a_sb, int64_t a_sn, int64_t a_sd,
                         const float *xyz2, int64_t b_sb, int64_t b_sn, int64_t b_sd,
                         int B, int N1, int N2, int D, float *out, rpe_stream_t stream);

/* ---- furthest_point_sampling -------------------------------------------------
 * Replaces furthest_point_sampling_kernel_wrapper(pts, dists_tmp, B, N, S, idx)
 * (furthest_point_sampling.cpp:3; kernel furthest_point_sampling_kernel.cu:34-85)
 * with the fallback's rules (wrapper.py:83-96): start at index 0,
 * nd = fl(fl(dx*dx+dy*dy)+dz*dz), running min, next = FIRST maximum.
 * xyz[b][n][d] = xyz[b*sb + n*sn + d*sd]; idx [B,S] int64 contiguous.
 * No scratch buffer: running distances live in registers.  Limits: N <= 32768. */
int rpe_fps(const float *xyz, int64_t sb, int64_t sn, int64_t sd,
            int B, int N, int S, int64_t *idx, rpe_stream_t stream);

/* ---- correlation2d forward ---------------------------------------------------
 * Replaces correlation_forward_kernel_wrapper(out, in1, in2, B, C, H, W, md)
 * (correlation.cpp:3-4; kernel correlation_forward_kernel.cu:11-54) but consumes
 * NCHW directly (the reference permutes both inputs to NHWC first, wrapper.py:68-69):
 *   out[b][(dy+md)*(2md+1)+(dx+md)][y][x] = (1/C) sum_c in1[b][c][y][x]*in2[b][c][y+dy][x+dx]
 * zero outside the image.  in1,in2 [B,C,H,W] contiguous; out [B,(2md+1)^2,H,W].
 * leaky_slope != 0 fuses the caller's leaky_relu (RPEFlow_core.py:362); pass 0
 * for the plain operator.  algo: 0 = pick, 1 = direct (any md), 2 = MFMA (md==4). */
int rpe_correlation2d_forward(const float *in1, const float *in2, int B, int C, int H, int W, int md,
                              float leaky_slope, int algo, float *out, rpe_stream_t stream);

/* ---- diagnostics -------------------------------------------------------------
 * Writes the lane/register map of v_mfma_f32_4x4x1_16b_f32 the correlation
 * kernel relies on: out[64*4] = D for A[lane]=lane, B[lane]=100*lane (one K).   */
int rpe_probe_mfma4x4(float *out256, rpe_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RPEFLOW_HIP_H */
