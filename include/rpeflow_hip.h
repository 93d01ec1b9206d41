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

#define RPE_ABI_VERSION 9

#define RPE_EINVAL (-1)       /* bad size / null pointer */
#define RPE_EUNSUPPORTED (-2) /* valid request this build has no kernel for */

typedef void *rpe_stream_t; /* a hipStream_t; NULL = the legacy default stream */

/* RPE_ABI_VERSION in the lower 16 bits.  The upper 16 are 0 in every library that computes right results; a diagnostic build
 * whose kernels drop work on purpose (energy probes, tools/corr_energy_probes.sh) reports its variant there, and
 * rpeflow_amd/_lib.py refuses to load it unless RPE_ALLOW_DIAGNOSTIC_LIB=1 says the caller expects one. */
int rpe_abi_version(void);
const char *rpe_error_string(int code);

/* ---- k_nearest_neighbor ---------------------------------------------------
 * Replaces k_nearest_neighbor_{2d,3d}_kernel_wrapper(b, n, m, k, query, input, idx)
 * (k_nearest_neighbor.cpp:3-4, kernels k_nearest_neighbor_kernel.cu:8-112) with
 * the arithmetic of the CPU fallback (wrapper.py:40-52,115-117):
 *   d = fl(fl(-2*dot + |q|^2) + |p|^2), dot = fma(q2,p2,fma(q1,p1,q0*p0)),
 * k smallest, ascending.  `mode` = one RPE_KNN_TIES_* value (what happens to EQUAL distances: the reference is matmul +
 * torch.topk on the CPU, i.e. libstdc++'s partial_sort / nth_element + sort, restated in knn.hip; k <= 63), optionally OR-ed
 * with one RPE_KNN_ALGO_* flag:
 *   RPE_KNN_TIES_TORCH  any tie among the k+1 best makes the query redo its selection exactly as torch.topk does it:
 *                       indices equal the reference's position for position;
 *   RPE_KNN_TIES_SET    only a tie between the k-th and the (k+1)-th distance triggers the redo: the returned neighbour
 *                       SET still always equals the reference's, equal distances inside the top k stay in index order;
 *   RPE_KNN_TIES_INDEX  lowest index first everywhere (no redo).
 * Either point layout is accepted through strides (elements, not bytes):
 *   input[b][m][d] = input[b*in_sb + m*in_sn + d*in_sd], same for query,
 * so channel-first callers need no transpose (wrapper.py:119-122 does one).
 * idx  [B,Q,k] int64 contiguous; dist [B,Q,k] fp32 contiguous or NULL.
 * Limits: 1 <= D <= 3, 1 <= k <= 64, k <= M (k = 64: index order whatever the mode).
 *
 * `workspace`: optional scratch (NULL, or 16-byte aligned device memory of at least rpe_knn_workspace_bytes(B, M, Q, D, k, mode)
 * bytes; uninitialised, free again when the call's kernels have run).  Results never depend on it; with it
 *   - k = 1, D = 2 searches of large clouds (the nearest projected point of every pixel, RPEFlow_core.py:327-330) run on a
 *     cloud binned into a uniform cell grid (csrc/knn_binned.hip: a wave's queries meet only the points of the cells around
 *     them; exact for any queries, fast for spatially coherent query order such as a raster): two launches;
 *   - the tied rows of a large k >= 2 search are redone by a second launch spread over the whole chip instead of by the waves
 *     that found them at the end of the first (csrc/knn.hip, knn_tie_replay_kernel).
 * RPE_KNN_ALGO_SWEEP: every query against every point whatever the sizes; RPE_KNN_ALGO_BINNED: the binned search or an error
 * (RPE_EUNSUPPORTED unless k = 1, D = 2, M >= 64; RPE_EINVAL without enough workspace); RPE_KNN_ALGO_MATRIX: the matrix-core
 * kernel wherever its structure allows (k <= 63, 64 k <= M, M >= 256) instead of only above the measured break-even sizes;
 * RPE_KNN_ALGO_INSERT: never the matrix-core kernel -- all four for cross-checks and for the break-even table
 * (tools/knn_gate_table.py); results are identical whichever kernel runs.                                                  */
#define RPE_KNN_TIES_INDEX 0
#define RPE_KNN_TIES_SET 1
#define RPE_KNN_TIES_TORCH 3
#define RPE_KNN_ALGO_SWEEP 0x100
#define RPE_KNN_ALGO_BINNED 0x200
#define RPE_KNN_ALGO_MATRIX 0x400
#define RPE_KNN_ALGO_INSERT 0x800
int64_t rpe_knn_workspace_bytes(int B, int M, int Q, int D, int k, int mode);
int rpe_knn(const float *input, int64_t in_sb, int64_t in_sn, int64_t in_sd,
            const float *query, int64_t q_sb, int64_t q_sn, int64_t q_sd,
            int B, int M, int Q, int D, int k, int mode,
            int64_t *idx, float *dist, void *workspace, int64_t workspace_bytes, rpe_stream_t stream);

/* Several searches with the same (B, D, k) in ONE call: job i is rpe_knn(jobs[i]...) exactly; jobs of one kind share a launch.
 * The PointConv pyramid's per-level searches (pointconv.py:46) depend on the sampled coordinates only and are issued together.
 * workspace: the jobs take rpe_knn_workspace_bytes(B, M_i, Q_i, D, k, mode) bytes each, in job order (pass the sum).       */
#define RPE_KNN_MAX_JOBS 8
typedef struct rpe_knn_job {
    const float *input;
    int64_t in_sb, in_sn, in_sd;   /* element strides: batch, point, dimension */
    const float *query;
    int64_t q_sb, q_sn, q_sd;
    int M, Q;
    int64_t *idx;                   /* [B,Q,k] */
    float *dist;                    /* [B,Q,k] or NULL */
} rpe_knn_job;
int rpe_knn_multi(const rpe_knn_job *jobs, int njobs, int B, int D, int k, int mode, void *workspace, int64_t workspace_bytes,
                  rpe_stream_t stream);

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
/* The same with the kernel chosen by the caller (identical indices either way; tests cross-check them):
 * RPE_FPS_PLAIN recomputes every point per sample, RPE_FPS_PRUNED skips Morton clusters the new sample cannot reach
 * (1024 < N <= 16384), RPE_FPS_AUTO = rpe_fps's choice (pruned for >= 2048 samples of >= 8192 points).        */
#define RPE_FPS_AUTO 0
#define RPE_FPS_PLAIN 1
#define RPE_FPS_PRUNED 2
int rpe_fps_algo(const float *xyz, int64_t sb, int64_t sn, int64_t sd,
                 int B, int N, int S, int64_t *idx, int algo, rpe_stream_t stream);

/* ---- correlation2d forward ---------------------------------------------------
 * Replaces correlation_forward_kernel_wrapper(out, in1, in2, B, C, H, W, md)
 * (correlation.cpp:3-4; kernel correlation_forward_kernel.cu:11-54) but consumes
 * NCHW directly (the reference permutes both inputs to NHWC first, wrapper.py:68-69):
 *   out[b][(dy+md)*(2md+1)+(dx+md)][y][x] = (1/C) sum_c in1[b][c][y][x]*in2[b][c][y+dy][x+dx]
 * zero outside the image.  in1,in2 [B,C,H,W] contiguous; out [B,(2md+1)^2,H,W].
 * leaky_slope != 0 fuses the caller's leaky_relu (RPEFlow_core.py:362); pass 0
 * for the plain operator.  algo: 0 = pick; 1 = direct (any md); 3 = small maps (md <= 4: 64 flattened pixels x one displacement
 * row a workgroup, channels split over its waves); 2 = MFMA tiles, register-staged (md == 4);
 * 7 / 8 = MFMA tiles behind an LDS-DMA ring, eight waves a workgroup with two rows / one row a wave (md == 4, W % 4 == 0,
 * C % 2 / C % 4 == 0, 16-byte aligned inputs; 8: the mid-size maps, where two rows a wave leave SIMDs empty).  All give the
 * same values to fp32 re-association; tests cross-check them.                                                        */
int rpe_correlation2d_forward(const float *in1, const float *in2, int B, int C, int H, int W, int md,
                              float leaky_slope, int algo, float *out, rpe_stream_t stream);

/* ---- correlation2d backward (training path of the operator) -------------------
 * Replaces correlation_backward_kernel_wrapper (correlation.cpp; kernels correlation_backward_kernel.cu:4-88)
 * behind CorrelationFunction.backward (wrapper.py:27-37), NCHW throughout:
 *   grad_in1[b][c][p] = 1/C sum_k grad_out[b][k][p]       in2[b][c][p + d_k]
 *   grad_in2[b][c][q] = 1/C sum_k grad_out[b][k][q - d_k] in1[b][c][q - d_k]      d_k = (k/(2md+1) - md, k%(2md+1) - md)
 * grad_out [B,(2md+1)^2,H,W]; either gradient pointer may be NULL to skip it.  md <= 4.                     */
int rpe_correlation2d_backward(const float *grad_out, const float *in1, const float *in2, int B, int C, int H, int W,
                               int md, float *grad_in1, float *grad_in2, rpe_stream_t stream);

/* ---- batch_indexing_channel_first / _last (models/utils.py:119-137, 101-116) ---
 * out[b][c][i] = data[b][c][idx[b][i]]   (data[b][c][n] = data[b*sb + c*sc + n*sn], out [B,C,I] contiguous)
 * out[b][i][c] = data[b][idx[b][i]][c]   (data[b][n][c] = data[b*sb + n*sn + c*sc], out [B,I,C] contiguous)
 * idx [B,I] int64 contiguous (callers flatten [B,I1,..,Im]).                     */
int rpe_gather_channel_first(const float *data, int64_t sb, int64_t sc, int64_t sn, const int64_t *idx,
                             int B, int C, int N, int I, float *out, rpe_stream_t stream);
int rpe_gather_channel_last(const float *data, int64_t sb, int64_t sn, int64_t sc, const int64_t *idx,
                            int B, int C, int N, int I, float *out, rpe_stream_t stream);

/* rpe_pointwise_conv: a stride-1 1x1 convolution with its epilogue in one launch (csrc/pointwise.hip; models/utils.py:7-62,
 * models/restormer_arch.py:88-110):  y[b][o][p] = act(scale[o] * sum_c W[o][c] x[b][c][p] + shift[o]) (+ residual[b][o][p]).
 * x [B,Cin,P] with batch stride x_batch_stride floats (Cin * P for a dense tensor; larger for a channel slice of a wider one),
 * y [B,Cout,P] contiguous fp32, residual [B,Cout,P] with batch stride residual_batch_stride (>= Cout * P: a channel slice of a wider
 * tensor is added where it lies); packed_weight [ceil(Cout/16)][ceil(Cin/4)][64]: entry (ot, kt, 16 k + i)
 * = W[16 ot + i][4 kt + k], zero outside (the MFMA A-fragment order); weight_batch_stride 0: one weight for the batch, > 0: ONE
 * PACKED WEIGHT PER SAMPLE, W[b] at packed_weight + b * weight_batch_stride -- the channel attention's per-sample matrix M[b]
 * (rpe_channel_attention_matrix, packed = 1) applied to v[b] with the block's residual in the epilogue; scale / shift /
 * residual may be NULL; act 0 none, 1 relu, 2 leaky_relu(slope).  Meant for the latency-bound layers (below ~0.5 GFLOP);
 * deterministic.                                                                                                       */
int rpe_pointwise_conv(const float *x, int64_t x_batch_stride, int B, int Cin, int64_t P, const float *packed_weight,
                       int64_t weight_batch_stride, int Cout, const float *scale, const float *shift, int act, float slope,
                       const float *residual, int64_t residual_batch_stride, float *y, rpe_stream_t stream);
/* rpe_im2col: cols [B, C*kh*kw, Ho*Wo] = torch.nn.functional.unfold(x [B,C,H,W], (kh,kw), dilation, padding, stride) for the whole
 * batch in one launch: the input side of the small convolutions that run as one deterministic GEMM instead of MIOpen's
 * atomically accumulating split-K kernels (rpeflow_amd/utils.py, wants_im2col) -- with the per-channel epilogue of the layer
 * that produced x applied to every value read: act(in_scale[c] * x + in_shift[c]), act 0 none / 1 ReLU / 2 LeakyReLU(in_slope),
 * padding stays 0, so that a run of such convolutions (the context network's dilated layers, the coarsest level's estimators)
 * needs no pass of its own for bias / BatchNorm / activation between two layers.  in_scale / in_shift may be NULL (1 / 0).  */
int rpe_im2col(const float *x, int B, int C, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
               const float *in_scale, const float *in_shift, int in_act, float in_slope, float *cols, rpe_stream_t stream);

/* ---- knn_interpolation after its KNN (models/utils.py:148-154) ------------------
 * w_j = 1/max(||in_xyz[:,knn_j] - q_xyz||_2, 1e-8), normalised over the k neighbours;
 * out[b][c][q] = sum_j (scale*F[b][c][knn_j]) * w_j.  scale = -1 gives backwarp_3d's
 * "-flow12" features (utils.py:166-167) without a negation pass.
 * F: channels [0, C) from `feat`, [C, C + C_b) from `feat_b` (either may be NULL with a count of 0) -- the decoder interpolates the coarser
 * level's [flow | flow features] (RPEFlow_core.py:352), two tensors that are never concatenated here.
 * `residual` [B, C + C_b, Q] through strides, or NULL: out = fl(residual + sum) (backwarp_3d's "xyz2 + flow21", utils.py:169).
 * Point/feature tensors channel-first through strides (sb, channel stride, point stride);
 * knn [B,Q,*] int64 with row stride knn_row_stride >= k; out [B, C + C_b, Q] contiguous. k <= 8. */
int rpe_knn_interpolate(const float *in_xyz, int64_t x_sb, int64_t x_sd, int64_t x_sn,
                        const float *feat, int64_t f_sb, int64_t f_sc, int64_t f_sn, int C,
                        const float *feat_b, int64_t g_sb, int64_t g_sc, int64_t g_sn, int C_b,
                        const float *q_xyz, int64_t q_sb, int64_t q_sd, int64_t q_sn,
                        const int64_t *knn, int64_t knn_row_stride,
                        int B, int M, int Q, int k, float scale,
                        const float *residual, int64_t r_sb, int64_t r_sc, int64_t r_sn, float *out, rpe_stream_t stream);

/* ---- bilinear sampling: backwarp_2d and grid_sample_wrapper ---------------------
 * (models/utils.py:186-198 and 288-294, i.e. F.grid_sample(bilinear, align_corners=True)
 * after the callers' 2*g/(S-1)-1 normalisation, restated as in ATen's CPU kernel.)
 * The sampled map is the channel-wise concatenation of 1..RPE_SAMPLE_MAX_SOURCES tensors of one H x W, which is never built
 * (`sources` is a HOST array): source i = channels planes data[b*sb + c*sc + pixel].  What the 3-D correlation fuser does
 * around its two grid_sample_wrapper calls (RPEFlow_core.py:105-111) rides in the launch:
 *   scale_even / scale_odd  every tap of an even / odd channel of the source is multiplied as it is read, fl(tap * s): sampling
 *                           "flow_2d * (sx, sy)" (:103-104) without that tensor; 1, 1 for a plain source;
 *   div_even / div_odd      (ABI 9) != 1: the product is then DIVIDED, fl(fl(tap * s) / d) -- the reference writes
 *                           "flow * (sensor_w - 1) / (image_w - 1)", a multiply and a divide, two roundings; with s = the numerator
 *                           and d = the denominator the taps carry the reference's bits (a single multiply by s / d can be an ulp off);
 *   subtract                NULL, or [B, channels, P] through strides: taken off the source's samples (":110  -= flow_3d[:, :2]").
 * Coordinates xy[b][d][p] = xy[b*xy_sb + d*xy_sd + p*xy_sp], d=0:x, 1:y.
 * add_pixel_grid=1, border=1, P=H*W: backwarp_2d(map, flow=xy, 'border').
 * add_pixel_grid=0, border=0:         grid_sample_wrapper(map, xy) (padding 'zeros').
 * out [B, sum channels, P] contiguous.                                              */
#define RPE_SAMPLE_MAX_SOURCES 4
typedef struct {
    const float *data;
    int64_t sb, sc;
    int channels;
    float scale_even, scale_odd;
    float div_even, div_odd;
    const float *subtract;
    int64_t sub_sb, sub_sc, sub_sp;
} rpe_sample_source;
int rpe_bilinear_sample(const rpe_sample_source *sources, int n_sources, int B, int H, int W,
                        const float *xy, int64_t xy_sb, int64_t xy_sd, int64_t xy_sp, int P,
                        int add_pixel_grid, int border, float *out, rpe_stream_t stream);
/* rpe_project_points: project_pc2image (models/utils.py:260-285) followed by the sensor -> feature-map rescale of
 *   RPEFlow_core.py:316-324, for the clouds of BOTH frames in one launch (the reference: per frame two adds -- or a div, two muls
 *   and two adds --, a cat and two in-place muls).  xyz_a, xyz_b [B,3,N] through strides (batch, channel, point); xyz_b NULL:
 *   one tensor.  intrinsics NULL: 'parallel' projection, u = fl(fl(x + cx) * scale_x); else 'perspective' with per-sample
 *   (f, cx, cy) at intrinsics[b*intr_sb + 0..2]: u = fl(fl(cx_b + fl(fl(f_b / z) * x)) * scale_x); v likewise.
 *   out [2B (or B), 2, N] contiguous: the first tensor's samples, then the second's.                             */
int rpe_project_points(const float *xyz_a, int64_t a_sb, int64_t a_sd, int64_t a_sn,
                       const float *xyz_b, int64_t b_sb, int64_t b_sd, int64_t b_sn, int B, int N,
                       const float *intrinsics, int64_t intr_sb, float cx, float cy, float scale_x, float scale_y,
                       float *out, rpe_stream_t stream);
/* rpe_resize_frames: RPEFlow.forward's input preparation (models/RPEFlow.py:40-47, utils.py:227-241): src [B,C,H,W]
 *   (uint8 when src_is_u8, else float) resized to [Ho,Wo] with F.interpolate(bilinear, align_corners=True); each tap is
 *   divided by ``divisor`` first when divisor != 0 (the images' / 255).  pair_split: C = 2*c channels are the two frames of
 *   a pair and land stacked on the batch axis, out [2B,c,Ho,Wo] (frame 1 of every sample, then frame 2); else out
 *   [B,C,Ho,Wo].                                                                                              */
int rpe_resize_frames(const void *src, int src_is_u8, float divisor, int pair_split, int B, int C, int H, int W,
                      int Ho, int Wo, float *out, rpe_stream_t stream);
/* rpe_resize_flow2d: the forward's last step (models/utils.py:217-224, called from RPEFlow.py:95): flow [B,2,H,W] resized to
 *   [Ho,Wo] with F.interpolate(bilinear, align_corners=True), channel 0 then times scale_x (= Wo / W rounded to fp32),
 *   channel 1 times scale_y, in one launch.  out [B,2,Ho,Wo] contiguous.                                         */
int rpe_resize_flow2d(const float *flow, int B, int H, int W, int Ho, int Wo, float scale_x, float scale_y,
                      float *out, rpe_stream_t stream);
/* rpe_upsample2x_pair: the 2-D decoder's coarse-to-fine hand-over (models/RPEFlow_core.py:364-369): a [B,Ca,h,w] (times
 *   scale_a) and b [B,Cb,h,w], both F.interpolate(scale_factor=2, mode='bilinear', align_corners=True), in one launch.
 *   out_a [B,Ca,2h,2w], out_b [B,Cb,2h,2w] contiguous.  Either tensor may be absent (C = 0, pointers NULL).        */
int rpe_upsample2x_pair(const float *a, int Ca, float scale_a, const float *b, int Cb, int B, int h, int w,
                        float *out_a, float *out_b, rpe_stream_t stream);

/* ---- IDS transforms of the clouds (models/utils.py:320-377; RPEFlow.py:68-69, 88-93) ---------------------------
 * rpe_ids_forward: perspect2parallel for n_clouds clouds stored as channel triples of pcs [B,3*n_clouds,N]
 *   (pcs[b][c][n] = pcs[b*sb + c*sc + n*sn]); intrinsics[b] = (f, cx, cy) at intrinsics[b*i_sb + 0..2];
 *   x' = (cx + f/z*x)*sw - hw,  y' = (cy + f/z*y)*sh - hh,  z' = (f*log z + 1)*sz, every operation rounded to fp32
 *   as the reference's CPU path rounds it, log = the correctly rounded fp32 logarithm.  sw = (Wp-1)/(W-1),
 *   sh = (Hp-1)/(H-1), hw = (Wp-1)/2, hh = (Hp-1)/2, sz = min(sw, sh), rounded to fp32 by the caller.
 *   out [n_clouds*B,3,N] contiguous, cloud-major: torch.cat([pc1, pc2], dim=0) of RPEFlow.py:68-69 + pwc3d_core.py:12.
 * rpe_ids_flow_inverse: out = parallel2perspect(xyz + flow) - parallel2perspect(xyz)   (RPEFlow.py:91-93),
 *   xyz, flow [B,3,N] through strides (batch, channel, point), out [B,3,N] contiguous.                        */
int rpe_ids_forward(const float *pcs, int64_t sb, int64_t sc, int64_t sn, const float *intrinsics, int64_t i_sb,
                    int B, int n_clouds, int N, float sw, float sh, float hw, float hh, float sz,
                    float *out, rpe_stream_t stream);
int rpe_ids_flow_inverse(const float *xyz, int64_t x_sb, int64_t x_sc, int64_t x_sn,
                         const float *flow, int64_t f_sb, int64_t f_sc, int64_t f_sn,
                         const float *intrinsics, int64_t i_sb, int B, int N,
                         float sw, float sh, float hw, float hh, float sz, float *out, rpe_stream_t stream);

/* ---- project_feat_with_nn_corr (models/utils.py:297-317) -------------------------
 * For pixel p with nearest projected point i = nn_idx[b][p]:
 *   out[b][0:2][p] = xy[b][:,i] - (p%W, p/W);  out[b][2][p] = mean_c(sample(feat_2d, xy_i)[c]*feat_2d[b][c][p]);
 *   out[b][3+c][p] = feat_3d[b][c][i].   feat_2d [B,C2,H,W] contiguous, out [B,C3+3,H,W].
 * workspace: B*N*(round4(C2)+round4(C3)) floats of scratch, 16-byte aligned (per-point rows: the samples are taken once per
 * point, as the reference does, then gathered per pixel; round4(n) = n rounded up to a multiple of 4).
 * sampled_2d [B,C2,N] through element strides (batch, channel, point), or NULL: what grid_sample_wrapper(feat_2d, xy) returned --
 *   the 3-D fuser of the same (map, points) pair computes it anyway (RPEFlow_core.py:334-337, 394-395), and the per-point
 *   bilinear taps are the expensive half of this operator.
 * The two element-wise steps the 2-D correlation fuser puts behind the operator (RPEFlow_core.py:82-83) ride in the launch:
 *   `subtract` [B, n_subtract, H*W] is subtracted from the LAST n_subtract projected channels ("projected 3-D flow minus the
 *   2-D flow") and `append` [B, n_append, H*W] is copied behind the C3 + 3 channels (the cat with the event features):
 *   out [B, C3 + 3 + n_append, H, W].  Either may be NULL with a count of 0.
 * feat_3d as two tensors, C3 = C3a + C3b: channels [0, C3a) from feat_3d, [C3a, C3) from feat_3d_b (NULL with C3b = 0), whose
 *   even / odd channels are multiplied by scale_even / scale_odd and, where div_even / div_odd != 1, then divided by them as they
 *   are read, fl(fl(v * s) / d) -- the 2-D correlation fuser projects [3-D cost volume | xy of the 3-D flow in feature-map units]
 *   (RPEFlow_core.py:363-366, 371-373: "flow * (image_w - 1) / (sensor_w - 1)", a multiply, a divide and a cat there).  */
int rpe_project_feat_nn_corr(const float *xy, int64_t xy_sb, int64_t xy_sd, int64_t xy_sn, const float *feat_2d, int C2,
                             int H, int W, const float *sampled_2d, int64_t sm_sb, int64_t sm_sc, int64_t sm_sn,
                             const float *feat_3d, int64_t f3_sb, int64_t f3_sc, int64_t f3_sn, int C3a,
                             const float *feat_3d_b, int64_t g3_sb, int64_t g3_sc, int64_t g3_sn, int C3b,
                             float scale_even, float scale_odd, float div_even, float div_odd,
                             const int64_t *nn_idx, const float *subtract, int n_subtract, const float *append, int n_append,
                             int B, int N, float *workspace, float *out, rpe_stream_t stream);

/* ---- PointConv in one kernel (models/pointconv.py:33-61, 90-122) -----------------
 * rpe_pointconv_pack_rows: rows[b][m][:] = [xyz[b][:,m] | srcs[0][b][:,m] | ... | zeros] -- cat([xyz, features]) channel-last
 *   (pointconv.py:43-44) together with the caller's own concatenation of up to four feature tensors in front of it
 *   (RPEFlow_core.py:382-391).  xyz [B,3,M] and every source [B,C_i,M] through element strides (batch, channel, point;
 *   src_strides = 3 per source); rows [B,M,CFp] contiguous, CFp a multiple of 16 with 3 + sum C_i <= CFp.
 *   srcs / src_strides / src_channels are HOST arrays.
 * rpe_pointconv_fused: from rows, the neighbour table and the query coordinates to the layer's output:
 *   G[q][w][c] = sum_{j<16} wn_j[w] * rows[b][knn[b][q][j]][c],   wn_j = leaky(W2 leaky(W1 (rows[knn_j][0:3] - q_xyz[:,q]) + b1) + b2)
 *   out[q][o]  = act(scale[o] * sum_{w,c} L[o][w*CF + c] G[q][w][c] + shift[o])
 *   with L = nn.Linear's weight (pointconv.py:13, flattened weight-major :57) given PRE-PACKED in MFMA fragment order:
 *   packed_linear[ci][g][t][kk][n][s] = L[16t + n][(4kk + s)*CF + 16ci + g], zero outside, ci < CFp/16, g < 16,
 *   t < n_tiles (a multiple of 8 with 16*n_tiles >= Cout), kk < 4, n < 16, s < 4.
 *   scale / shift [Cout] or NULL fold bias and eval-mode BatchNorm (pointconv.py:58-59); act 0 none, 1 relu, 2 leaky_relu.
 *   out_mode 0: out [B,Cout,Q] channel-first (what the reference returns); out_mode 1: out [B,Q,out_stride] in the rows
 *   format above ([q_xyz | y | zeros]), i.e. the next PointConvNoSampling's input on the same points.
 *   Nothing of size [B,Q,16,CF] or [B,Q,16*CF] is written to memory.  Limits: k = 16, M*CFp < 2^31.                */
int rpe_pointconv_pack_rows(const float *xyz, int64_t x_sb, int64_t x_sd, int64_t x_sn,
                            const float *const *srcs, const int64_t *src_strides, const int *src_channels, int n_src,
                            int B, int M, int CFp, float *rows, rpe_stream_t stream);
int rpe_pointconv_fused(const float *rows, int CFp, int M, const int64_t *knn, int64_t knn_row_stride,
                        const float *q_xyz, int64_t q_sb, int64_t q_sd, int64_t q_sn,
                        const float *w1, const float *b1, const float *w2, const float *b2, float leaky_slope,
                        const float *packed_linear, int n_tiles, const float *scale, const float *shift,
                        int act, float act_slope, int B, int Q, int Cout, int out_mode, int out_stride,
                        float *out, rpe_stream_t stream);

/* ---- point-wise MLP: one or two Conv1dNormRelu layers (models/utils.py:7-98) in one launch -------------------------
 * y = act2(s2 * (W2 act1(s1 * (W1 x) + t1)) + t2) per point; x [B,C0,N] through element strides (batch, channel, point).
 * Layer widths are padded to 16 * T1 / 16 * T2 (T2 = 0: one layer).  Packed operands (built once per module):
 *   w1_packed [ceil(C0/16)][T1][4][16][4]: [g][t][kk][o][s] = W1[16t + o][16g + 4kk + s]   (zero beyond C1 x C0)
 *   w2_packed [T1][T2][4][16][4]:          [t1][t2][kk][o][s] = W2[16t2 + o][16t1 + 4kk + s]
 *   scale_shift{1,2} [2][16 T]: the per-channel scale then shift that bias and eval-mode BatchNorm fold into
 *   act 0 none, 1 relu, 2 leaky_relu(slope).
 * out_mode 0: out [B,Cout,N] channel-first; out_mode 1: out [B,N,out_stride] = [xyz | y | zeros], the PointConv
 * kernel's gather source (xyz [B,3,N] through strides).  Supported (T1,T2): (1,1) (1,2) (2,4) (4,6) (6,8) (8,12) (8,4)
 * and (T,0) for T in 1, 2, 4, 6, 8, 12; anything else returns RPE_EUNSUPPORTED.                                     */
int rpe_mlp1d_fused(const float *x, int64_t x_sb, int64_t x_sc, int64_t x_sn, int B, int C0, int N,
                    const float *w1_packed, const float *scale_shift1, int act1, int T1,
                    const float *w2_packed, const float *scale_shift2, int act2, int T2,
                    float slope, int Cout, int out_mode, int out_stride,
                    const float *xyz, int64_t z_sb, int64_t z_sd, int64_t z_sn,
                    float *out, rpe_stream_t stream);

/* ---- Correlation3D (models/pwc3d_core.py:69-117) behind its neighbour search, two launches -----------------------
 * Channel counts are padded to Cp = 16*T, T in {1,2,4,6,8,12}; every [Cp] / packed array below is zero beyond C.
 * rpe_corr3d_cost: the point-to-neighbour cost p2n (:84-98) for every point n of cloud 1, neighbours knn[b][n][0..15] in cloud 2:
 *     hidden[j][c] = leaky(p1_rows[b][n][c] + p2_rows[b][knn_j][c] + wc4[c][0..2] . rel_j),   rel_j = xyz_s[:,knn_j] - xyz_q[:,n]
 *     cost[j][c']  = leaky(sum_c W2[c'][c] hidden[j][c] + b2[c'])
 *     p2n[n][c']   = sum_j relu(W3 relu(Wn2 relu(Wn1 rel_j + bn1) + bn2) + b3)[c'] * cost[j][c']
 *   p1_rows [B,N,Cp] = Wa feat1 + bias and p2_rows [B,M,Cp] = Wb feat2, channel-last, are the per-POINT halves of cost_mlp's
 *   first 1x1 conv (its input is the concatenation [feat1 | feat2_nbr | rel], :92-94; one small GEMM each, left to the
 *   caller); wc4 [Cp,4] = that conv's last three input columns (4th float unused).
 *   w2_packed [T][T][4][16][4]: [g][t][kk][n][s] = W2[16t + n][16g + 4kk + s] (cost_mlp's second conv), b2 [Cp].
 *   Weight net (weight_net2 here, weight_net1 in rpe_corr3d_n2n; MLP2d(3,[8,8,C],relu), :66-67): n_w1 [8,3], n_b1 [8],
 *   n_w2 [8,8], n_b2 [8], n_w3_packed [T][4][16][2]: [t][kk][n][s] = W3[16t + n][4s + kk], n_b3 [Cp].
 *   p2n_rows [B,N,Cp] channel-last (the second hop gathers whole rows).
 * rpe_corr3d_n2n: out[b][c][n] = sum_j weight_net1(xyz[:,knn_j] - xyz[:,n])[c] * p2n_rows[b][knn_j][c]   (:106-115),
 *   knn = the neighbours of cloud 1 in itself; out [B,C,N] channel-first (what the reference returns).
 * xyz_* channel-first through element strides (batch, dim, point); knn [B,N,*] int64, row stride >= 16.               */
int rpe_corr3d_cost(const float *p1_rows, const float *p2_rows, const float *wc4, const float *w2_packed, const float *b2,
                    const float *n_w1, const float *n_b1, const float *n_w2, const float *n_b2,
                    const float *n_w3_packed, const float *n_b3,
                    const float *xyz_q, int64_t q_sb, int64_t q_sd, int64_t q_sn,
                    const float *xyz_s, int64_t s_sb, int64_t s_sd, int64_t s_sn,
                    const int64_t *knn, int64_t knn_row_stride, int B, int Cp, int N, int M,
                    float leaky_slope, float *p2n_rows, rpe_stream_t stream);
int rpe_corr3d_n2n(const float *p2n_rows, const float *n_w1, const float *n_b1, const float *n_w2, const float *n_b2,
                   const float *n_w3_packed, const float *n_b3,
                   const float *xyz, int64_t x_sb, int64_t x_sd, int64_t x_sn,
                   const int64_t *knn, int64_t knn_row_stride, int B, int C, int Cp, int N,
                   float *out, rpe_stream_t stream);

/* ---- Restormer-block pieces (models/restormer_arch.py) -------
 * rpe_dwconv3: depth-wise convolution, stride 1, zero padding 1: kh = 3 -> 3x3 over [B,C,H,W] (weight
 *   [C,1,3,3]); kh = 1 -> 3-tap over [B,C,W] with H = 1 (weight [C,1,3]).  The input is the channel
 *   concatenation of up to three contiguous tensors in0 [B,C0,H,W], in1 [B,C1,H,W], in2 [B,C2,H,W]
 *   (C1 = C2 = 0 for one tensor): qkv_dwconv(torch.cat((x, y, y))) of :182/:263 without the cat.
 *   bias [C] or NULL.  gate = 1: out[b][c] = gelu(conv[c]) * conv[c + C/2] for c < C/2 (the GDFN
 *   gate, :104-105/:244-245; erf GELU), out [B,C/2,H,W]; gate = 0: out [B,C,H,W].
 * rpe_channel_layernorm: out[b][c][p] = (x - mean_c) / sqrt(var_c + eps) * weight[c] + bias[c] over the
 *   channel axis of [B,C,P] (biased variance; WithBias_LayerNorm :47-63); bias = NULL gives BiasFree
 *   (:31-44: x / sqrt(var + eps) * weight, no mean subtraction).                                  */
int rpe_dwconv3(const float *in0, int C0, const float *in1, int C1, const float *in2, int C2,
                const float *weight, const float *bias, int B, int H, int W, int kh, int gate,
                float *out, rpe_stream_t stream);
/* (x1 .. out1: a second tensor of the same shape with its own affine parameters in the same launch -- norm1x(x), norm1y(y) of
 * the cross blocks, restormer_arch.py:218, 298 -- or NULL)                                                                  */
int rpe_channel_layernorm(const float *x0, const float *weight0, const float *bias0, float *out0,
                          const float *x1, const float *weight1, const float *bias1, float *out1,
                          int B, int C, int64_t P, float eps, rpe_stream_t stream);
/* rpe_gdfn_tail: the gated feed-forward behind project_in in ONE launch (restormer_arch.py:104-106, 244-246): depth-wise 3x3
 *   (kh = 3; t [B, 2*hidden, H, W]) or 3-tap (kh = 1, H = 1; t [B, 2*hidden, W]) convolution with dw_weight [2*hidden, kh*3] and
 *   dw_bias [2*hidden] or NULL, g = gelu(first half) * second half, then project_out: y[b][o][p] = sum_c Wout[o][c] g[b][c][p] +
 *   shift[o] (+ residual[b][o][p]; y may BE residual).  packed_weight: Wout [Cout, hidden] in rpe_pointwise_conv's fragment order.
 *   The arithmetic of rpe_dwconv3(gate = 1) followed by rpe_pointwise_conv, without the gated tensor's round trip through HBM.
 *   Needs W % 4 == 0, Cout <= 128, 16-byte aligned t / y / residual: else RPE_EUNSUPPORTED (callers keep the two launches).
 *   Faster than the two launches for the 3-tap form (the point-cloud blocks: 20-27 against 32 us); for 3x3 maps the gate is three
 *   times the work and the two launches win (level 1, C = 96: 237 against 296 us): rpeflow_amd uses it for kh = 1 only.       */
int rpe_gdfn_tail(const float *t, int B, int hidden, int H, int W, int kh, const float *dw_weight, const float *dw_bias,
                  const float *packed_weight, int Cout, const float *shift, const float *residual, float *y, rpe_stream_t stream);
/* rpe_channel_attention_matrix: the attention core of Mutual_Attention{2D,3D}.forward (restormer_arch.py:184-203,
 *   265-282) up to and including project_out, as a per-batch C x C matrix (C = heads * c):
 *     attn_h = softmax_j( normalize(q_h) normalize(k_h)^T * temperature[h] ),  F.normalize over the P positions (eps),
 *     m_out[b] = w_out blockdiag_h(attn_h)                                      w_out [C,C] = project_out.weight
 *   so that project_out(attn v) = m_out[b] v[b] -- the caller finishes with one batched GEMM (+ residual).
 *   q, k: [B][C][P] fp32 with row stride P and the given batch stride in floats (views into the qkv tensor).
 *   c <= 96.  workspace: rpe_channel_attention_workspace_floats(B, heads, c, P) floats.                         */
int64_t rpe_channel_attention_workspace_floats(int B, int heads, int c, int64_t P);
/*   packed = 1: every m_out[b] in rpe_pointwise_conv's weight-fragment order ([ceil(C/16)][ceil(C/4)][64] floats per sample, zero
 *   outside): the block then ends with ONE rpe_pointwise_conv launch, out = residual + m[b] v[b].                            */
int rpe_channel_attention_matrix(const float *q, const float *k, int64_t batch_stride, const float *temperature,
                                 const float *w_out, int B, int heads, int c, int64_t P, float eps,
                                 float *workspace, float *m_out, int packed, rpe_stream_t stream);
/* rpe_convex_upsample: RAFT-style convex up-sampling of a 2-D flow (models/utils.py:201-214):
 *   out[b][c][h*s+i][w*s+j] = sum_k softmax_k(mask[b][k*s*s + i*s + j][h][w]) * s * flow[b][c][h + k/3 - 1][w + k%3 - 1]
 *   (zero outside).  flow [B,2,H,W], mask [B,9*s*s,H,W], out [B,2,H*s,W*s]; s in {2,4,8}.                        */
int rpe_convex_upsample(const float *flow, const float *mask, int B, int H, int W, int scale, float *out, rpe_stream_t stream);
/* rpe_events_to_voxel: event stream -> voxel grid with temporal bilinear weights (event_utils.py:109-128, 211-303).
 *   Events are given stably sorted by pixel (pixel = y * W + x): pixel_sorted, t_sorted (the raw timestamps: float64, or
 *   float32 with t_is_f32 -- what load_events_h5 returns, event_utils.py:11-20; the arithmetic is then float32 throughout,
 *   as numpy and torch do it on such arrays), polarity_sorted, and the first index of every run of equal pixels
 *   (run_start [runs]).  t_first / t_last: timestamps of the first and last event of the ORIGINAL order.  out [C][HW] fp32,
 *   zero-initialised by the caller; C = bins, or 2 * bins with split_polarity (positive grids, then negative grids).  Sums
 *   are formed per pixel in the events' original order: bit-identical to the reference's CPU index_put_(accumulate=True). */
int rpe_events_to_voxel(const int *pixel_sorted, const void *t_sorted, int t_is_f32, const int *polarity_sorted,
                        const int *run_start, int runs, int n_events, double t_first, double t_last, int bins,
                        int split_polarity, int64_t HW, float *out, rpe_stream_t stream);
/* rpe_channel_affine_act: y[b][c][p] = act(scale[c]*y[b][c][p] + shift[c]) IN PLACE over [B,C,P] -- the bias add,
 *   eval-mode BatchNorm and activation after a Conv{1,2}dNormRelu convolution (models/utils.py:7-62) in one pass.
 *   scale / shift may be NULL (1 / 0).  act: 0 none, 1 relu, 2 leaky_relu(slope).                              */
int rpe_channel_affine_act(float *y, const float *scale, const float *shift, int B, int C, int64_t P,
                           int act, float slope, rpe_stream_t stream);
/* rpe_channel_affine_add_act: y = act(scale[c]*y + shift[c] + zscale[c]*z) IN PLACE on y, z [B,C,P] -- the tail of the 2-D
 *   pyramid's residual block, act(BN(conv1(.)) + BN(down0(x))) (pwc2d_core.py:6-25), over the two raw convolution outputs in
 *   one pass.  scale / shift / zscale may be NULL (1 / 0 / 1); shift carries both branches' shifts.                */
int rpe_channel_affine_add_act(float *y, const float *scale, const float *shift, const float *z, const float *zscale, int B, int C,
                               int64_t P, int act, float slope, rpe_stream_t stream);
/* rpe_residual_tail: the tail of the 2-D pyramid's residual block with its shortcut branch inside (pwc2d_core.py:6-25): in place on y [B][Cout][Ho][Wo]
 * (the raw output of the block's second convolution), y = act(scale * y + shift + shortcut_scale * (W0 . x[:, :, ::stride, ::stride]))
 * with x [B][Cin][H][W] the block's input, W0 [Cout][Cin] the 1x1 shortcut convolution (no padding; Ho = (H - 1) / stride + 1),
 * shift the sum of both branches' shifts; act 0 none / 1 ReLU / 2 LeakyReLU(slope).  Cin <= 256. */
int rpe_residual_tail(float *y, const float *scale, const float *shift, const float *x, const float *shortcut_weight,
                              const float *shortcut_scale, int B, int Cin, int Cout, int H, int W, int stride, int act, float slope,
                              rpe_stream_t stream);

/* ---- evaluation metric sums (eval_withocc.py:65-108, eval_noocc.py:57-99) ----------------------------------
 * The caller of the hot path: per batch, the twelve sums the reference's Evaluator keeps (there: a Python loop over
 * samples with ~12 .item() host syncs each).  acc[12] float64, ADDED to:
 *   [0..3]  2-D: #valid pixels, sum EPE, #(EPE < 1 px), #(EPE > 3 and EPE/|gt| > 0.05)
 *   [4..7]  3-D: #valid points, sum EPE, #(EPE < 0.05), #(EPE < 0.1)
 *   [8..11] the 3-D group again over valid points with occ_mask == 0 (occ_mask NULL: untouched)
 * flow2d [B,2,HW], target2d [B,c2,HW] (c2 = 3: channel 2 is the validity mask, > 0 valid), flow3d [B,3,N], target3d
 * [B,c3,N] (c3 = 4: channel 3 is the mask), occ_mask [B,N] or NULL; all contiguous fp32.  EPE = sqrt(sum diff^2) in fp32,
 * NaN EPEs are invalid; sums in float64, per-block partials in `workspace` (rpe_eval_workspace_doubles doubles) added in
 * a fixed order: no atomics.                                                                                       */
int rpe_eval_workspace_doubles(int64_t n_pixels, int64_t n_points);
int rpe_eval_accumulate(const float *flow2d, const float *target2d, int target2d_channels, int B, int64_t HW,
                        const float *flow3d, const float *target3d, int target3d_channels, int64_t N, const float *occ_mask,
                        double *workspace, double *acc, rpe_stream_t stream);

/* ---- diagnostics -------------------------------------------------------------
 * rpe_clock_stamp: when the stream reaches this point one thread stores slot2[0] = the engine cycle counter (s_memtime) and
 *   slot2[1] = the constant-rate counter (s_memrealtime, wall_clock64: 100 MHz).  Usable inside a captured HIP graph: the
 *   constant-rate entries are the timelines of multi-stream replays that rocprofv3 serialises (rpeflow_amd.model.StampTrace,
 *   tools/stamp_timeline.py), the kernel is the marker tools/trace_window.py looks for.  The cycle counters of two stamps must
 *   NOT be subtracted: they may come from different XCDs / shader engines, whose counters are not one clock.
 * rpe_clock_stamp_all: the same pair of counters stored per COMPUTE UNIT -- slots[2 key], slots[2 key + 1], key = XCC_ID << 8 |
 *   HW_ID[15:8] (shader engine, array, CU), 2048 keys: `slots` is 4096 values, 16-byte aligned, zeroed by the caller (a pair is ONE
 *   16-byte store: both counters of a slot come from one wave); 8192 one-wave workgroups cover the chip.  Two of these bracket a stretch of the stream; for every key both reached, d(cycles) / d(ticks) x the
 *   constant rate = the clock that compute unit ran at (rpeflow_amd.runtime.ShaderClock takes the median; bench.py's
 *   roofline_corr carries it).  wall_khz (host pointer, may be NULL): that rate on the current device
 *   (hipDeviceAttributeWallClockRate).                                                                              */
int rpe_clock_stamp(unsigned long long *slot2, rpe_stream_t stream);
int rpe_clock_stamp_all(unsigned long long *slots, int *wall_khz, rpe_stream_t stream);

#ifdef RPE_EXPERIMENTAL /* only in a library built with -DRPE_EXPERIMENTAL (python -m rpeflow_amd.build --experimental) */
/* Writes the lane/register map of v_mfma_f32_4x4x1_16b_f32 the correlation
 * kernel relies on: out[64*4] = D for A[lane]=lane, B[lane]=100*lane (one K).   */
int rpe_probe_mfma4x4(float *out256, rpe_stream_t stream);
#endif

#ifdef __cplusplus
}
#endif
#endif /* RPEFLOW_HIP_H */
