// Gather / interpolation / sampling glue of models/utils.py as single kernels.
//
// The reference composes each of these from 3-10 PyTorch kernels with expanded
// int64 index tensors (utils.py:133-135) and a normalise/denormalise round trip
// through F.grid_sample (utils.py:186-198, 288-294).  Here every op is one launch,
// indices are read once per output row, and lanes always walk the contiguous
// output dimension so stores are coalesced; gathered reads hit L2 (the sources
// are a few MB at most).
#include <math.h>


#include "common.h"

namespace {

// out[b][c][i] = data[b][c][idx[b][i]]           (batch_indexing_channel_first, utils.py:119-137)
__global__ __launch_bounds__(256) void gather_cf_kernel(const float *__restrict__ data, int64_t sb, int64_t sc, int64_t sn,
                                                        const int64_t *__restrict__ idx, int C, int N, int I, int c_per_block,
                                                        float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z;
    if (i >= I) return;
    const int64_t src = idx[(int64_t)b * I + i];
    const int c0 = blockIdx.y * c_per_block, c1 = min(C, c0 + c_per_block);
    const float *d = data + (int64_t)b * sb + src * sn;
    float *o = out + ((int64_t)b * C) * I + i;
    for (int c = c0; c < c1; ++c) o[(int64_t)c * I] = d[(int64_t)c * sc];
}

// out[b][i][c] = data[b][idx[b][i]][c]           (batch_indexing_channel_last, utils.py:101-116)
__global__ __launch_bounds__(256) void gather_cl_kernel(const float *__restrict__ data, int64_t sb, int64_t sn, int64_t sc,
                                                        const int64_t *__restrict__ idx, int C, int I,
                                                        float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (e >= (int64_t)I * C) return;
    const int i = (int)(e / C), c = (int)(e % C);
    const int64_t src = idx[(int64_t)b * I + i];
    out[(int64_t)b * I * C + e] = data[(int64_t)b * sb + src * sn + (int64_t)c * sc];
}

// knn_interpolation (utils.py:140-156) after the KNN: inverse-distance weights of the
// k neighbours, w_j = 1/max(||p_j - q||_2, 1e-8), normalised, weighted feature sum.
// Features: channels [0, Ca) from `feat`, [Ca, C) from `feat_b` (the decoder interpolates [flow | flow features] of the coarser
// level, RPEFlow_core.py:352: two tensors, never concatenated); `residual` (or NULL): added to the result, fl(residual + sum)
// (backwarp_3d's "xyz2 + flow21", utils.py:169).
template <int KMAX>
__global__ __launch_bounds__(256) void knn_interp_kernel(const float *__restrict__ in_xyz, int64_t x_sb, int64_t x_sd, int64_t x_sn,
                                                         const float *__restrict__ feat, int64_t f_sb, int64_t f_sc, int64_t f_sn, int Ca,
                                                         const float *__restrict__ feat_b, int64_t g_sb, int64_t g_sc, int64_t g_sn,
                                                         const float *__restrict__ q_xyz, int64_t q_sb, int64_t q_sd, int64_t q_sn,
                                                         const int64_t *__restrict__ knn, int64_t k_sq, int k, int C, int Q,
                                                         float negate, int cpb, const float *__restrict__ residual, int64_t r_sb, int64_t r_sc,
                                                         int64_t r_sn, float *__restrict__ out) {
    // a thread: one query, cpb channels (blockIdx.y): with one thread per query and all channels in its loop the small
    // levels ran 256-1024 threads on the whole chip, 15-25 us of dependent gathers for a few MB
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z;
    if (q >= Q) return;
    const float *xb = in_xyz + (int64_t)b * x_sb, *qb = q_xyz + (int64_t)b * q_sb;
    const float qx = qb[q * q_sn], qy = qb[q_sd + q * q_sn], qz = qb[2 * q_sd + q * q_sn];
    int64_t id[KMAX];
    float w[KMAX];
    float wsum = 0.f;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        if (j < k) {
            id[j] = knn[((int64_t)b * Q + q) * k_sq + j];
            const float dx = xb[id[j] * x_sn] - qx, dy = xb[x_sd + id[j] * x_sn] - qy, dz = xb[2 * x_sd + id[j] * x_sn] - qz;
            float d = sqrtf(dx * dx + dy * dy + dz * dz);
            d = fmaxf(d, 1e-8f);
            w[j] = 1.0f / d;
            wsum += w[j];
        }
    }
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
        if (j < k) w[j] = w[j] / wsum;
    const float *fa = feat + (int64_t)b * f_sb, *fg = feat_b ? feat_b + (int64_t)b * g_sb : nullptr;
    const int c0 = blockIdx.y * cpb, c1 = min(C, c0 + cpb);
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const bool first = c < Ca;
        const float *plane = first ? fa + (int64_t)c * f_sc : fg + (int64_t)(c - Ca) * g_sc;
        const int64_t sn = first ? f_sn : g_sn;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if (j < k) s += (negate * plane[id[j] * sn]) * w[j];
        if (residual) s = residual[(int64_t)b * r_sb + (int64_t)c * r_sc + (int64_t)q * r_sn] + s;
        out[((int64_t)b * C + c) * Q + q] = s;
    }
}

// ATen CPU grid_sample un-normalisation after the callers' own normalisation:
//   gn = 2*g/(S-1) - 1 (utils.py:189-190, 290-291);  u = (gn+1) * ((S-1)/2);  border: clamp to [0,S-1]
__device__ __forceinline__ float unnormalise(float g, int S, bool border) {
    const float gn = 2.0f * g / (float)(S - 1) - 1.0f;
    float u = (gn + 1.0f) * ((float)(S - 1) / 2.0f);
    if (border) {
        u = (u > 0.0f) ? u : 0.0f;  // NaN -> 0 like clamp_min's operand order
        u = fminf(u, (float)(S - 1));
    }
    return u;
}

struct Bilinear {
    int o_nw, o_ne, o_sw, o_se;  // offsets inside one H*W plane, -1 = outside the image
    float w_nw, w_ne, w_sw, w_se;
    __device__ __forceinline__ void setup(float gx, float gy, int H, int W, bool border) {
        const float u = unnormalise(gx, W, border), v = unnormalise(gy, H, border);
        const float fx = floorf(u), fy = floorf(v);
        const float w = u - fx, e = 1.0f - w, n = v - fy, s = 1.0f - n;
        w_nw = s * e; w_ne = s * w; w_sw = n * e; w_se = n * w;
        const bool ok_x = fx >= -2.0f && fx <= (float)W + 1.0f, ok_y = fy >= -2.0f && fy <= (float)H + 1.0f;
        const int ix = ok_x ? (int)fx : -2, iy = ok_y ? (int)fy : -2;
        const bool in_w = ix >= 0 && ix < W, in_e = ix + 1 >= 0 && ix + 1 < W;
        const bool in_n = iy >= 0 && iy < H, in_s = iy + 1 >= 0 && iy + 1 < H;
        o_nw = (in_n && in_w) ? iy * W + ix : -1;
        o_ne = (in_n && in_e) ? iy * W + ix + 1 : -1;
        o_sw = (in_s && in_w) ? (iy + 1) * W + ix : -1;
        o_se = (in_s && in_e) ? (iy + 1) * W + ix + 1 : -1;
    }
    __device__ __forceinline__ float sample(const float *plane) const {
        const float v_nw = o_nw >= 0 ? plane[o_nw] : 0.f, v_ne = o_ne >= 0 ? plane[o_ne] : 0.f;
        const float v_sw = o_sw >= 0 ? plane[o_sw] : 0.f, v_se = o_se >= 0 ? plane[o_se] : 0.f;
        float acc = v_nw * w_nw + v_ne * w_ne;
        acc = acc + v_sw * w_sw;
        acc = acc + v_se * w_se;
        return acc;
    }
    // the same on the map (plane * scale / div): every tap rounded once (twice with a divisor) more, as reading the caller's
    // "x * num / den" tensor would (RPEFlow_core.py:367-370)
    __device__ __forceinline__ float sample_scaled(const float *plane, float scale, float div) const {
        const float v_nw = o_nw >= 0 ? rpe_scaled(plane[o_nw], scale, div) : 0.f, v_ne = o_ne >= 0 ? rpe_scaled(plane[o_ne], scale, div) : 0.f;
        const float v_sw = o_sw >= 0 ? rpe_scaled(plane[o_sw], scale, div) : 0.f, v_se = o_se >= 0 ? rpe_scaled(plane[o_se], scale, div) : 0.f;
        float acc = v_nw * w_nw + v_ne * w_ne;
        acc = acc + v_sw * w_sw;
        acc = acc + v_se * w_se;
        return acc;
    }
};

// Workgroup -> (channel block, position block) for the kernels that GATHER from channel planes at scattered positions.
// Workgroups are dealt round-robin over the 8 XCDs (ids i and i + 8 share one, each with its own 4 MB L2).  With the position
// block as the fastest index every XCD met every plane: at level 1 (81 planes of 138 KB per sample, 4096 scattered points) each
// XCD's 512 points touch ~60 % of a plane's cache lines, so the fabric carried 8 x 0.6 = 4.8x the map (point_rows_kernel: 45 us
// for 45 MB).  Here XCD x owns the channel blocks cb = x (mod 8): a plane crosses the fabric once.  grid.x = xcd_grid(ncb, npb).
__device__ __forceinline__ bool xcd_block(int ncb, int &cb, int &pb) {
    const int per = (ncb + 7) >> 3;
    const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    cb = (t % per) * 8 + xcd;
    pb = t / per;
    return cb < ncb;
}
static inline unsigned xcd_grid(int ncb, int npb) { return (unsigned)(((ncb + 7) >> 3) * 8 * npb); }

// out[b][c][p] = bilinear(map[b][c], (gx,gy)[b][p]);  add_grid: coordinates are
// pixel (p % W, p / W) + flow  (backwarp_2d, utils.py:186-198); otherwise xy as given
// (grid_sample_wrapper, utils.py:288-294).
// The map is the channel-wise concatenation of up to RPE_SAMPLE_MAX_SOURCES tensors of one H x W (rpe_sample_source), which is
// never built: the 3-D correlation fuser samples [cost volume | flow in sensor units] and the event features at the same points
// and concatenates the results (RPEFlow_core.py:105-111).  A source's taps are multiplied by its per-channel-parity scale as
// they are read -- fl(tap * s), what sampling the caller's "flow * scale" tensor reads -- and `subtract` comes off its result.
struct SampleSources {
    rpe_sample_source src[RPE_SAMPLE_MAX_SOURCES];
    int first[RPE_SAMPLE_MAX_SOURCES + 1];  // first output channel of every source; first[n] = C
    int n;
};

__global__ __launch_bounds__(256) void bilinear_kernel(SampleSources S, int C, int H, int W,
                                                       const float *__restrict__ xy, int64_t xy_sb, int64_t xy_sd, int64_t xy_sp,
                                                       int P, int add_grid, int border, int c_per_block, int xcd_map,
                                                       float *__restrict__ out) {
    // scattered positions (grid_sample_wrapper): the XCD-aware mapping above; a warp (adjacent pixels sample adjacent taps: a
    // plane's lines are met once anyway) keeps the plain 3-D grid, which measured faster there (65 against 97 us at level 1)
    int cb = blockIdx.y, pb = blockIdx.x;
    if (xcd_map && !xcd_block((C + c_per_block - 1) / c_per_block, cb, pb)) return;
    const int p = pb * blockDim.x + threadIdx.x;
    const int b = blockIdx.z;
    if (p >= P) return;
    float gx = xy[(int64_t)b * xy_sb + (int64_t)p * xy_sp];
    float gy = xy[(int64_t)b * xy_sb + xy_sd + (int64_t)p * xy_sp];
    if (add_grid) {
        gx = (float)(p % W) + gx;
        gy = (float)(p / W) + gy;
    }
    Bilinear bl;
    bl.setup(gx, gy, H, W, border != 0);
    const int c0 = cb * c_per_block, c1 = min(C, c0 + c_per_block);
    int c = c0;
    while (c < c1) {
        int si = 0;
        while (si + 1 < S.n && c >= S.first[si + 1]) ++si;  // (wave-uniform: c is)
        const rpe_sample_source &src = S.src[si];
        const int ce = min(c1, S.first[si + 1]);
        const float *base = src.data + (int64_t)b * src.sb + (int64_t)(c - S.first[si]) * src.sc;
        const bool plain = src.scale_even == 1.0f && src.scale_odd == 1.0f && src.div_even == 1.0f && src.div_odd == 1.0f && !src.subtract;
        if (plain) {
            // four channels per trip: 16 independent gathered loads in flight instead of 4
            for (; c + 4 <= ce; c += 4, base += 4 * src.sc) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = bl.sample(base + (int64_t)u * src.sc);
#pragma unroll
                for (int u = 0; u < 4; ++u) out[((int64_t)b * C + c + u) * P + p] = v[u];
            }
            for (; c < ce; ++c, base += src.sc) out[((int64_t)b * C + c) * P + p] = bl.sample(base);
        } else {
            for (; c < ce; ++c, base += src.sc) {
                const int lc = c - S.first[si];
                float v = bl.sample_scaled(base, (lc & 1) ? src.scale_odd : src.scale_even, (lc & 1) ? src.div_odd : src.div_even);
                if (src.subtract) v = v - src.subtract[(int64_t)b * src.sub_sb + (int64_t)lc * src.sub_sc + (int64_t)p * src.sub_sp];
                out[((int64_t)b * C + c) * P + p] = v;
            }
        }
    }
}

// Coarse-to-fine hand-over of the 2-D decoder (RPEFlow_core.py:364-369 / pwc2d_core usage): the coarser level's flow (times
// `scale_a`, 2 in the model) and flow features, both up-sampled x2 with F.interpolate(bilinear, align_corners=True), in ONE
// launch instead of a multiply and two interpolations.  Arithmetic as ATen's upsample_bilinear2d: source = dst * (in-1)/(out-1),
// h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11).  Channels of a (Ca) then b (Cb) form one virtual channel axis.
__global__ __launch_bounds__(256) void upsample2x_pair_kernel(const float *__restrict__ a, int Ca, float scale_a, const float *__restrict__ b,
                                                              int Cb, int h, int w, int c_per_block, float *__restrict__ out_a,
                                                              float *__restrict__ out_b) {
    const int H = 2 * h, W = 2 * w, P = H * W;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.z;
    if (p >= P) return;
    const int y = p / W, x = p - y * W;
    const float ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float sy = ry * (float)y, sx = rx * (float)x;
    const int y0 = (int)sy, x0 = (int)sx;
    const int yp = y0 < h - 1 ? 1 : 0, xp = x0 < w - 1 ? 1 : 0;
    const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
    const int o00 = y0 * w + x0, o01 = o00 + xp, o10 = o00 + yp * w, o11 = o10 + xp;
    const int hw = h * w;
    const int c0 = blockIdx.y * c_per_block, c1 = min(Ca + Cb, c0 + c_per_block);
    for (int c = c0; c < c1; ++c) {
        const bool first = c < Ca;
        const float *src = first ? a + ((int64_t)n * Ca + c) * hw : b + ((int64_t)n * Cb + (c - Ca)) * hw;
        const float k = first ? scale_a : 1.f;
        const float v00 = src[o00] * k, v01 = src[o01] * k, v10 = src[o10] * k, v11 = src[o11] * k;
        const float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
        float *dst = first ? out_a + ((int64_t)n * Ca + c) * P : out_b + ((int64_t)n * Cb + (c - Ca)) * P;
        dst[p] = v;
    }
}

// Input preparation of RPEFlow.forward (models/RPEFlow.py:40-47 with utils.py:227-241): the frames resized to multiples
// of 64 (F.interpolate bilinear, align_corners=True) in one pass -- for the uint8 image pair the conversion to float, the
// division by 255 and the split into frame 1 / frame 2 stacked on the batch axis ([B,6,H,W] -> [2B,3,Ho,Wo]) are folded in.
// Per tap v = float(u8) / 255 (a true division, as the reference's), then ATen's h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11).
template <typename T>
__global__ __launch_bounds__(256) void resize_frames_kernel(const T *__restrict__ src, int B, int C, int H, int W, int Ho, int Wo, float divisor,
                                                            int pair_split, int c_per_block, float post_x, float post_y, int post,
                                                            float *__restrict__ out) {
    const int P = Ho * Wo;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.z;
    if (p >= P) return;
    const int y = p / Wo, x = p - y * Wo;
    const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const float sy = ry * (float)y, sx = rx * (float)x;
    const int y0 = (int)sy, x0 = (int)sx;
    const int yp = y0 < H - 1 ? 1 : 0, xp = x0 < W - 1 ? 1 : 0;
    const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
    const int o00 = y0 * W + x0, o01 = o00 + xp, o10 = o00 + yp * W, o11 = o10 + xp;
    const int64_t hw = (int64_t)H * W;
    const int c0 = blockIdx.y * c_per_block, c1 = min(C, c0 + c_per_block);
    const int half = C / 2;
    for (int c = c0; c < c1; ++c) {
        const T *s = src + ((int64_t)n * C + c) * hw;
        float v00 = (float)s[o00], v01 = (float)s[o01], v10 = (float)s[o10], v11 = (float)s[o11];
        if (divisor != 0.f) { v00 = v00 / divisor; v01 = v01 / divisor; v10 = v10 / divisor; v11 = v11 / divisor; }
        float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
        if (post) v = v * (c == 0 ? post_x : post_y);  // resize_flow2d: flow[:, 0] *= tw / w, flow[:, 1] *= th / h (utils.py:222-223)
        const int64_t plane = pair_split ? ((int64_t)(c / half) * B + n) * half + (c % half) : (int64_t)n * C + c;
        out[plane * P + p] = v;
    }
}

// project_feat_with_nn_corr (utils.py:297-317) in two launches.
//   point_rows_kernel:   rows[b][i][:] = [ sample(feat_2d[b], xy_i)[0..C2) | feat_3d[b][:, i] ]   (channel-last, per POINT)
//   project_rows_kernel: for pixel p with nearest point i = nn[b][p]:
//       out[0:2] = xy_i - pixel, out[2] = mean_c(rows[i][c] * feat_2d[c][p]), out[3:] = rows[i][C2:]
// The reference samples per point too (grid_sample_wrapper over all N points, utils.py:308) and then
// gathers per pixel; sampling per pixel instead would repeat the 4-corner fetch ~HW/N = 8 times.
// A pixel's gather is one contiguous row here instead of C2+C3 strided reads.
// feat_3d of project_feat_with_nn_corr as up to two tensors: channels [0, Ca) from `a`, [Ca, C3) from `b` times a per-channel-parity
// scale -- the 2-D correlation fuser projects [3-D cost volume | 3-D flow's xy in feature-map units] (RPEFlow_core.py:371-373: two
// in-place muls and a cat there), which is never concatenated here.
struct Feat3 {
    const float *a;
    int64_t a_sb, a_sc, a_sn;
    int Ca;
    const float *b;
    int64_t b_sb, b_sc, b_sn;
    float s_even, s_odd, d_even, d_odd;
    __device__ __forceinline__ float at(int batch, int c, int i) const {
        if (c < Ca) return a[(int64_t)batch * a_sb + (int64_t)c * a_sc + (int64_t)i * a_sn];
        const int lc = c - Ca;
        return rpe_scaled(b[(int64_t)batch * b_sb + (int64_t)lc * b_sc + (int64_t)i * b_sn], (lc & 1) ? s_odd : s_even, (lc & 1) ? d_odd : d_even);
    }
};

__global__ __launch_bounds__(256) void point_rows_kernel(const float *__restrict__ xy, int64_t xy_sb, int64_t xy_sd, int64_t xy_sn,
                                                         const float *__restrict__ feat2d, int C2, int H, int W, Feat3 F3,
                                                         int C3, int N, int c_per_block, const float *__restrict__ sampled,
                                                         int64_t sm_sb, int64_t sm_sc, int64_t sm_sn, float *__restrict__ rows) {
    const int CT = C2 + C3;
    int cb, pb;
    if (!xcd_block((CT + c_per_block - 1) / c_per_block, cb, pb)) return;
    const int i = pb * blockDim.x + threadIdx.x;
    const int b = blockIdx.z;
    if (i >= N) return;
    const int c0 = cb * c_per_block, c1 = min(CT, c0 + c_per_block);
    const int C2p = (C2 + 3) & ~3, RS = C2p + ((C3 + 3) & ~3);  // both halves of a row start on 16 bytes (float4 reads per pixel)
    float *row = rows + ((int64_t)b * N + i) * RS;
    if (c0 < C2 && sampled) {  // the caller has grid_sample_wrapper(feat_2d, xy) already (the 3-D fuser of the same pair needs it)
        const float *sp = sampled + (int64_t)b * sm_sb + (int64_t)i * sm_sn;
        const int ce = min(c1, C2);
        int c = c0;
        for (; c + 8 <= ce; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sp[(int64_t)(c + u) * sm_sc];
#pragma unroll
            for (int u = 0; u < 8; ++u) row[c + u] = v[u];
        }
        for (; c < ce; ++c) row[c] = sp[(int64_t)c * sm_sc];
    } else if (c0 < C2) {
        const float px = xy[(int64_t)b * xy_sb + (int64_t)i * xy_sn], py = xy[(int64_t)b * xy_sb + xy_sd + (int64_t)i * xy_sn];
        Bilinear bl;
        bl.setup(px, py, H, W, false);
        const int64_t HW = (int64_t)H * W;
        const float *f2 = feat2d + (int64_t)b * C2 * HW;
        const int ce = min(c1, C2);
        int c = c0;
        for (; c + 4 <= ce; c += 4) {  // four channels per trip: 16 independent gathered loads in flight
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = bl.sample(f2 + (int64_t)(c + u) * HW);
#pragma unroll
            for (int u = 0; u < 4; ++u) row[c + u] = v[u];
        }
        for (; c < ce; ++c) row[c] = bl.sample(f2 + (int64_t)c * HW);
    }
    {
        float *row3 = row + C2p - C2;  // logical channel c >= C2 lives at row[C2p + (c - C2)]
        int c = max(c0, C2);
        for (; c + 8 <= c1; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = F3.at(b, c + u - C2, i);
#pragma unroll
            for (int u = 0; u < 8; ++u) row3[c + u] = v[u];
        }
        for (; c < c1; ++c) row3[c] = F3.at(b, c - C2, i);
    }
}

// point_rows_kernel's job when the caller brings the samples (the decoder levels do): a transpose of [C2 | C3] channel
// planes into per-point rows.  One thread per (point, channel slice) writes 32-byte pieces of 64 different rows; here a
// workgroup takes PB consecutive points, reads every plane's PB values as one coalesced segment, turns the tile in LDS
// (odd pitch: no bank conflicts) and writes the PB rows -- contiguous in memory -- as float4.
__global__ __launch_bounds__(256) void point_rows_copy_kernel(const float *__restrict__ sampled, int64_t sm_sb, int64_t sm_sc, int64_t sm_sn,
                                                              int C2, Feat3 F3, int C3, int N, int PB, float *__restrict__ rows) {
    extern __shared__ float tile[];
    const int C2p = (C2 + 3) & ~3, C3p = (C3 + 3) & ~3, RS = C2p + C3p, RP = RS | 1;
    const int b = blockIdx.y, i0 = blockIdx.x * PB;
    const int pl = threadIdx.x % PB, g = threadIdx.x / PB, G = 256 / PB;
    const int i = min(i0 + pl, N - 1);
    float *mine = tile + pl * RP;
    constexpr int CU = 16;  // (level 1: 118 planes over 8 slices = 15 loads a thread, one round trip)
    {
        const float *sp = sampled + (int64_t)b * sm_sb + (int64_t)i * sm_sn;
        for (int c = g; c < C2; c += G * CU) {
            float v[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) v[u] = sp[(int64_t)min(c + u * G, C2 - 1) * sm_sc];
#pragma unroll
            for (int u = 0; u < CU; ++u)
                if (c + u * G < C2) mine[c + u * G] = v[u];
        }
        if (g == 0)
            for (int c = C2; c < C2p; ++c) mine[c] = 0.f;
    }
    if (C3 > 0) {
        for (int c = g; c < C3; c += G * CU) {
            float v[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) v[u] = F3.at(b, min(c + u * G, C3 - 1), i);
#pragma unroll
            for (int u = 0; u < CU; ++u)
                if (c + u * G < C3) mine[C2p + c + u * G] = v[u];
        }
        if (g == 0)
            for (int c = C3; c < C3p; ++c) mine[C2p + c] = 0.f;
    }
    __syncthreads();
    const int npts = min(PB, N - i0), RS4 = RS >> 2;
    float4 *dst = reinterpret_cast<float4 *>(rows + ((int64_t)b * N + i0) * RS);
    for (int e = threadIdx.x; e < npts * RS4; e += 256) {
        const int r = e / RS4, q = (e - r * RS4) * 4;
        const float *t = tile + r * RP + q;
        dst[e] = make_float4(t[0], t[1], t[2], t[3]);
    }
}

__global__ __launch_bounds__(256) void project_rows_kernel(const float *__restrict__ xy, int64_t xy_sb, int64_t xy_sd, int64_t xy_sn,
                                                           const float *__restrict__ feat2d, int C2, int H, int W, int C3, int N,
                                                           const float *__restrict__ rows, const int64_t *__restrict__ nn,
                                                           const float *__restrict__ sub, int n_sub, const float *__restrict__ tail, int n_tail,
                                                           float *__restrict__ out) {
    // sub [B][n_sub][HW]: subtracted from the LAST n_sub projected channels (the 2-D correlation fuser's "projected flow minus
    // the 2-D flow", RPEFlow_core.py:82); tail [B][n_tail][HW]: copied behind the C3 + 3 channels (its cat with the event
    // features, :83): the fuser's MLP input leaves this kernel complete.
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    const int HW = H * W;
    if (p >= HW) return;
    const int64_t i = nn[(int64_t)b * HW + p];
    const float px = xy[(int64_t)b * xy_sb + i * xy_sn], py = xy[(int64_t)b * xy_sb + xy_sd + i * xy_sn];
    float *o = out + (int64_t)b * (C3 + 3 + n_tail) * HW + p;
    o[0] = px - (float)(p % W);
    o[HW] = py - (float)(p / W);
    // A pixel's row is a gather: 64 lanes, 64 rows.  Read as float4 (the two halves of a row start on 16 bytes) it is a quarter
    // of the load instructions (level 1: 19.5 -> 19.1 us, 70 MB: the kernel is at the bandwidth a copy reaches).
    const int C2p = (C2 + 3) & ~3, RS = C2p + ((C3 + 3) & ~3);
    const float *row = rows + ((int64_t)b * N + i) * RS;
    const float4 *row4 = reinterpret_cast<const float4 *>(row);
    const float *f2 = feat2d + (int64_t)b * C2 * HW + p;
    float s = 0.f;
    int c = 0;
    // sixteen (then four) channels in flight; the sum keeps the reference's channel order
    for (; c + 16 <= C2; c += 16) {
        float4 r[4];
        float f[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = row4[(c >> 2) + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) f[u] = f2[(int64_t)(c + u) * HW];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s += r[u].x * f[4 * u];
            s += r[u].y * f[4 * u + 1];
            s += r[u].z * f[4 * u + 2];
            s += r[u].w * f[4 * u + 3];
        }
    }
    for (; c + 4 <= C2; c += 4) {
        const float4 r = row4[c >> 2];
        float f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) f[u] = f2[(int64_t)(c + u) * HW];
        s += r.x * f[0];
        s += r.y * f[1];
        s += r.z * f[2];
        s += r.w * f[3];
    }
    for (; c < C2; ++c) s += row[c] * f2[(int64_t)c * HW];
    o[2 * (int64_t)HW] = s / (float)C2;
    const float *row3 = row + C2p;
    const float4 *row34 = reinterpret_cast<const float4 *>(row3);
    c = 0;
    for (; c + 16 <= C3; c += 16) {
        float4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = row34[(c >> 2) + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            o[(int64_t)(3 + c + 4 * u) * HW] = r[u].x;
            o[(int64_t)(3 + c + 4 * u + 1) * HW] = r[u].y;
            o[(int64_t)(3 + c + 4 * u + 2) * HW] = r[u].z;
            o[(int64_t)(3 + c + 4 * u + 3) * HW] = r[u].w;
        }
    }
    for (; c + 4 <= C3; c += 4) {
        const float4 r = row34[c >> 2];
        o[(int64_t)(3 + c) * HW] = r.x;
        o[(int64_t)(3 + c + 1) * HW] = r.y;
        o[(int64_t)(3 + c + 2) * HW] = r.z;
        o[(int64_t)(3 + c + 3) * HW] = r.w;
    }
    for (; c < C3; ++c) o[(int64_t)(3 + c) * HW] = row3[c];
    for (int t = 0; t < n_sub; ++t) {  // (after the copy above: same thread, same addresses)
        float *q = o + (int64_t)(3 + C3 - n_sub + t) * HW;
        *q = *q - sub[((int64_t)b * n_sub + t) * HW + p];
    }
    c = 0;
    const float *tp = tail + (int64_t)b * n_tail * HW + p;
    float *ot = o + (int64_t)(3 + C3) * HW;
    for (; c + 8 <= n_tail; c += 8) {
        float r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = tp[(int64_t)(c + u) * HW];
#pragma unroll
        for (int u = 0; u < 8; ++u) ot[(int64_t)(c + u) * HW] = r[u];
    }
    for (; c < n_tail; ++c) ot[(int64_t)c * HW] = tp[(int64_t)c * HW];
}

// The same on the coarse maps (a few hundred to a few thousand pixels): one thread per pixel is a handful of waves, each
// walking C2 + C3 channels as a chain of memory round trips (18 x 30: 18 us, four calls per level on the decoder's critical
// path).  A workgroup takes 64 pixels and splits the channels over its CG waves: every wave sums its slice of the correlation
// term and copies its slice of the projected and appended channels; the partial sums meet in LDS.
template <int CG>
__global__ __launch_bounds__(CG * 64) void project_rows_small_kernel(const float *__restrict__ xy, int64_t xy_sb, int64_t xy_sd, int64_t xy_sn,
                                                                     const float *__restrict__ feat2d, int C2, int H, int W, int C3, int N,
                                                                     const float *__restrict__ rows, const int64_t *__restrict__ nn,
                                                                     const float *__restrict__ sub, int n_sub, const float *__restrict__ tail,
                                                                     int n_tail, float *__restrict__ out) {
    __shared__ float red[CG][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int HW = H * W;
    const int p = blockIdx.x * 64 + lane, b = blockIdx.y;
    const bool valid = p < HW;
    const int pc = valid ? p : HW - 1;  // (lanes past the end read the last pixel and store nothing)
    const int64_t i = nn[(int64_t)b * HW + pc];
    const int C2p = (C2 + 3) & ~3, RS = C2p + ((C3 + 3) & ~3);
    const float *row = rows + ((int64_t)b * N + i) * RS;
    const float *f2 = feat2d + (int64_t)b * C2 * HW + pc;
    float *o = out + (int64_t)b * (C3 + 3 + n_tail) * HW + pc;
    constexpr int CU = 8;
    {  // this wave's slice of the correlation term
        const int per = (C2 + CG - 1) / CG, c0 = grp * per, c1 = min(C2, c0 + per);
        float s = 0.f;
        for (int c = c0; c < c1; c += CU) {
            float r[CU], f[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int cc = min(c + u, c1 - 1);
                r[u] = row[cc];
                f[u] = f2[(int64_t)cc * HW];
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) s += c + u < c1 ? r[u] * f[u] : 0.f;
        }
        red[grp][lane] = s;
    }
    {  // ... of the projected 3-D channels (the last n_sub of them minus `sub`)
        const int per = (C3 + CG - 1) / CG, c0 = grp * per, c1 = min(C3, c0 + per);
        const float *sp = n_sub > 0 ? sub + ((int64_t)b * n_sub - (C3 - n_sub)) * HW + pc : row;  // channel c of the projected ones: sp[c * HW], c >= C3 - n_sub
        for (int c = c0; c < c1; c += CU) {
            float r[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int cc = min(c + u, c1 - 1);
                r[u] = row[C2p + cc];
                if (cc >= C3 - n_sub) r[u] = r[u] - sp[(int64_t)cc * HW];
            }
#pragma unroll
            for (int u = 0; u < CU; ++u)
                if (valid && c + u < c1) o[(int64_t)(3 + c + u) * HW] = r[u];
        }
    }
    {  // ... of the appended channels
        const int per = (n_tail + CG - 1) / CG, c0 = grp * per, c1 = min(n_tail, c0 + per);
        const float *tp = tail + (int64_t)b * n_tail * HW + pc;
        float *ot = o + (int64_t)(3 + C3) * HW;
        for (int c = c0; c < c1; c += CU) {
            float r[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) r[u] = tp[(int64_t)min(c + u, c1 - 1) * HW];
#pragma unroll
            for (int u = 0; u < CU; ++u)
                if (valid && c + u < c1) ot[(int64_t)(c + u) * HW] = r[u];
        }
    }
    __syncthreads();
    if (grp == 0 && valid) {
        const float px = xy[(int64_t)b * xy_sb + i * xy_sn], py = xy[(int64_t)b * xy_sb + xy_sd + i * xy_sn];
        o[0] = px - (float)(p % W);
        o[HW] = py - (float)(p / W);
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < CG; ++g) s += red[g][lane];
        o[2 * (int64_t)HW] = s / (float)C2;
    }
}

int channel_split(int C, long items, int B);
int channel_split(int C, long items, int B) {
    // enough blocks to fill 256 CUs without making each thread's channel loop trivial
    long blocks = ((items + 255) / 256) * B;
    int split = 1;
    while (split < C && blocks * split < 2048 && C / (split * 2) >= 4) split *= 2;
    return (C + split - 1) / split;
}

}  // namespace

namespace {
// cols[b][(c*kh + i)*kw + j][oy*Wo + ox] = x[b][c][oy*sh - ph + i*dh][ox*sw - pw + j*dw] (0 outside): torch.nn.functional.unfold
// for the whole batch in one launch (ATen's im2col runs one kernel per sample).  A thread writes four consecutive output
// positions (one 16-byte store when Ho*Wo is a multiple of 4); lanes run along the positions, so for stride 1 the reads
// of a wave are contiguous row pieces of x.
// in_scale / in_shift / in_act: the per-channel epilogue act(scale * x + shift) of the layer that PRODUCED x, applied to the values
// read (padding stays 0): a run of im2col convolutions passes each layer's bias / BatchNorm / activation on to the next unfold
// instead of running a pass of its own over the raw GEMM output.
__global__ __launch_bounds__(256) void im2col_kernel(const float *__restrict__ x, int C, int H, int W, int kh, int kw, int sh, int sw, int ph,
                                                     int pw, int dh, int dw, int Ho, int Wo, const float *__restrict__ in_scale,
                                                     const float *__restrict__ in_shift, int in_act, float in_slope, float *__restrict__ cols) {
    const int64_t P = (int64_t)Ho * Wo;
    const int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int row = blockIdx.y, b = blockIdx.z;  // row = (c*kh + i)*kw + j
    if (p0 >= P) return;
    const int j = row % kw, i = (row / kw) % kh, c = row / (kw * kh);
    const float *plane = x + ((int64_t)b * C + c) * H * W;
    int oy = (int)(p0 / Wo), ox = (int)(p0 - (int64_t)oy * Wo);
    const float a = in_scale ? in_scale[c] : 1.0f, s = in_shift ? in_shift[c] : 0.0f;
    float v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int y = oy * sh - ph + i * dh, xx = ox * sw - pw + j * dw;
        const bool in = p0 + t < P && y >= 0 && y < H && xx >= 0 && xx < W;
        float u = in ? plane[(int64_t)y * W + xx] : 0.f;
        if (in_scale || in_shift || in_act) {
            u = a * u + s;
            u = in_act == 1 ? rpe_relu(u) : (in_act == 2 ? (u >= 0.f ? u : u * in_slope) : u);
            u = in ? u : 0.f;
        }
        v[t] = u;
        if (++ox == Wo) ox = 0, ++oy;
    }
    float *out = cols + ((int64_t)b * C * kh * kw + row) * P + p0;
    if (p0 + 4 <= P && (P & 3) == 0) {
        *reinterpret_cast<float4 *>(out) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (p0 + t < P) out[t] = v[t];
    }
}
}  // namespace

RPE_API int rpe_im2col(const float *x, int B, int C, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                       const float *in_scale, const float *in_shift, int in_act, float in_slope, float *cols, rpe_stream_t stream) {
    if (in_act < 0 || in_act > 2) return RPE_EINVAL;
    if (!x || !cols || B < 0 || C < 1 || H < 1 || W < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1 || ph < 0 || pw < 0 || dh < 1 || dw < 1)
        return RPE_EINVAL;
    const int Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1, Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
    if (Ho < 1 || Wo < 1) return RPE_EINVAL;
    if (B == 0) return 0;
    if (B > 65535 || (int64_t)C * kh * kw > 65535) return RPE_EUNSUPPORTED;
    dim3 grid((unsigned)(((int64_t)Ho * Wo + 1023) / 1024), (unsigned)(C * kh * kw), (unsigned)B);
    hipLaunchKernelGGL(im2col_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, Ho, Wo, in_scale, in_shift,
                       in_act, in_slope, cols);
    return rpe_launch_status();
}

RPE_API int rpe_gather_channel_first(const float *data, int64_t sb, int64_t sc, int64_t sn, const int64_t *idx, int B, int C,
                                     int N, int I, float *out, rpe_stream_t stream) {
    if (!data || !idx || !out || B < 0 || C < 0 || N <= 0 || I < 0) return RPE_EINVAL;
    if (B == 0 || C == 0 || I == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    const int cpb = channel_split(C, I, B);
    dim3 grid((I + 255) / 256, (C + cpb - 1) / cpb, B);
    hipLaunchKernelGGL(gather_cf_kernel, grid, dim3(256), 0, (hipStream_t)stream, data, sb, sc, sn, idx, C, N, I, cpb, out);
    return rpe_launch_status();
}

RPE_API int rpe_gather_channel_last(const float *data, int64_t sb, int64_t sn, int64_t sc, const int64_t *idx, int B, int C,
                                    int N, int I, float *out, rpe_stream_t stream) {
    if (!data || !idx || !out || B < 0 || C < 0 || N <= 0 || I < 0) return RPE_EINVAL;
    if (B == 0 || C == 0 || I == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    const int64_t total = (int64_t)I * C;
    dim3 grid((unsigned)((total + 255) / 256), B);
    hipLaunchKernelGGL(gather_cl_kernel, grid, dim3(256), 0, (hipStream_t)stream, data, sb, sn, sc, idx, C, I, out);
    return rpe_launch_status();
}

RPE_API int rpe_knn_interpolate(const float *in_xyz, int64_t x_sb, int64_t x_sd, int64_t x_sn, const float *feat, int64_t f_sb,
                                int64_t f_sc, int64_t f_sn, int C, const float *feat_b, int64_t g_sb, int64_t g_sc, int64_t g_sn, int C_b,
                                const float *q_xyz, int64_t q_sb, int64_t q_sd, int64_t q_sn, const int64_t *knn, int64_t knn_row_stride,
                                int B, int M, int Q, int k, float scale, const float *residual, int64_t r_sb, int64_t r_sc, int64_t r_sn,
                                float *out, rpe_stream_t stream) {
    if (!in_xyz || !q_xyz || !knn || !out || B < 0 || M <= 0 || Q < 0 || C < 0 || C_b < 0 || k < 1 || (C > 0 && !feat) || (C_b > 0 && !feat_b)) return RPE_EINVAL;
    if (k > 8) return RPE_EUNSUPPORTED;
    const int CT = C + C_b;
    if (B == 0 || Q == 0 || CT == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    const int cpb = channel_split(CT, Q, B);
    dim3 grid((Q + 255) / 256, (CT + cpb - 1) / cpb, B);
    hipStream_t st = (hipStream_t)stream;
    if (k <= 3)
        hipLaunchKernelGGL(knn_interp_kernel<3>, grid, dim3(256), 0, st, in_xyz, x_sb, x_sd, x_sn, feat, f_sb, f_sc, f_sn, C, C_b > 0 ? feat_b : nullptr,
                           g_sb, g_sc, g_sn, q_xyz, q_sb, q_sd, q_sn, knn, knn_row_stride, k, CT, Q, scale, cpb, residual, r_sb, r_sc, r_sn, out);
    else
        hipLaunchKernelGGL(knn_interp_kernel<8>, grid, dim3(256), 0, st, in_xyz, x_sb, x_sd, x_sn, feat, f_sb, f_sc, f_sn, C, C_b > 0 ? feat_b : nullptr,
                           g_sb, g_sc, g_sn, q_xyz, q_sb, q_sd, q_sn, knn, knn_row_stride, k, CT, Q, scale, cpb, residual, r_sb, r_sc, r_sn, out);
    return rpe_launch_status();
}

RPE_API int rpe_bilinear_sample(const rpe_sample_source *sources, int n_sources, int B, int H, int W, const float *xy, int64_t xy_sb,
                                int64_t xy_sd, int64_t xy_sp, int P, int add_pixel_grid, int border, float *out, rpe_stream_t stream) {
    if (!sources || n_sources < 1 || n_sources > RPE_SAMPLE_MAX_SOURCES || !xy || !out || B < 0 || H < 1 || W < 1 || P < 0) return RPE_EINVAL;
    if (add_pixel_grid && P != H * W) return RPE_EINVAL;
    SampleSources S{};
    S.n = n_sources;
    int C = 0;
    for (int i = 0; i < n_sources; ++i) {
        if (sources[i].channels < 0 || (sources[i].channels > 0 && !sources[i].data)) return RPE_EINVAL;
        S.src[i] = sources[i];
        S.first[i] = C;
        C += sources[i].channels;
    }
    for (int i = n_sources; i <= RPE_SAMPLE_MAX_SOURCES; ++i) S.first[i] = C;
    if (B == 0 || C == 0 || P == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    int cpb = channel_split(C, P, B);
    dim3 grid((P + 255) / 256, (C + cpb - 1) / cpb, B);
    int xcd_map = 0;
    if (C >= 8) {
        xcd_map = 1;
        // XCD-aware mapping: at least 8 channel blocks (every XCD busy), about as many as the split above wanted
        const int want = (C + cpb - 1) / cpb;
        const int ncb = want <= 8 ? 8 : (want + 7) / 8 * 8;
        cpb = (C + ncb - 1) / ncb;
        grid = dim3(xcd_grid((C + cpb - 1) / cpb, (P + 255) / 256), 1, B);
    }
    hipLaunchKernelGGL(bilinear_kernel, grid, dim3(256), 0, (hipStream_t)stream, S, C, H, W, xy, xy_sb, xy_sd, xy_sp, P,
                       add_pixel_grid, border, cpb, xcd_map, out);
    return rpe_launch_status();
}

// project_pc2image (utils.py:260-285) + the sensor -> feature-map rescale of RPEFlow_core.py:316-324 for both frames' clouds in one
// launch: out[(f * B + b)][0][i] = fl(fl(x + cx) * sx)  ('parallel': cx, cy scalars) or fl(fl(cx_b + fl(fl(f_b / z) * x)) * sx)
// ('perspective': per-sample f, cx, cy), y likewise.  The reference: 2 x (two adds or div / mul / add pairs, a cat, two in-place muls).
__global__ __launch_bounds__(256) void project_points_kernel(const float *__restrict__ xyz_a, int64_t a_sb, int64_t a_sd, int64_t a_sn,
                                                             const float *__restrict__ xyz_b, int64_t b_sb, int64_t b_sd, int64_t b_sn, int B,
                                                             int N, const float *__restrict__ intr, int64_t intr_sb, float cx, float cy, float sx,
                                                             float sy, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int cloud = blockIdx.y;  // frame-major: [0, B) the first tensor's samples, [B, 2B) the second's
    if (i >= N) return;
    const bool second = cloud >= B;
    const int b = second ? cloud - B : cloud;
    const float *p = second ? xyz_b + (int64_t)b * b_sb + (int64_t)i * b_sn : xyz_a + (int64_t)b * a_sb + (int64_t)i * a_sn;
    const int64_t sd = second ? b_sd : a_sd;
    const float x = p[0], y = p[sd];
    float u, v;
    if (intr) {
        const float f = intr[(int64_t)b * intr_sb], icx = intr[(int64_t)b * intr_sb + 1], icy = intr[(int64_t)b * intr_sb + 2];
        const float fz = f / p[2 * sd];
        u = icx + fz * x;
        v = icy + fz * y;
    } else {
        u = x + cx;
        v = y + cy;
    }
    out[((int64_t)cloud * 2) * N + i] = u * sx;
    out[((int64_t)cloud * 2 + 1) * N + i] = v * sy;
}

RPE_API int rpe_project_points(const float *xyz_a, int64_t a_sb, int64_t a_sd, int64_t a_sn, const float *xyz_b, int64_t b_sb, int64_t b_sd,
                               int64_t b_sn, int B, int N, const float *intrinsics, int64_t intr_sb, float cx, float cy, float scale_x,
                               float scale_y, float *out, rpe_stream_t stream) {
    if (!xyz_a || !out || B < 0 || N < 0) return RPE_EINVAL;
    if (B == 0 || N == 0) return 0;
    if (B > 32767) return RPE_EUNSUPPORTED;  // (both frames ride in grid.y)
    hipLaunchKernelGGL(project_points_kernel, dim3((N + 255) / 256, xyz_b ? 2 * B : B), dim3(256), 0, (hipStream_t)stream, xyz_a, a_sb, a_sd, a_sn,
                       xyz_b, b_sb, b_sd, b_sn, B, N, intrinsics, intr_sb, cx, cy, scale_x, scale_y, out);
    return rpe_launch_status();
}

RPE_API int rpe_resize_frames(const void *src, int src_is_u8, float divisor, int pair_split, int B, int C, int H, int W, int Ho, int Wo,
                              float *out, rpe_stream_t stream) {
    if (!src || !out || B < 0 || C < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1 || (pair_split && (C & 1))) return RPE_EINVAL;
    if (B == 0) return 0;
    if (B > 65535 || (int64_t)Ho * Wo >= (1ll << 31) || (int64_t)H * W >= (1ll << 31)) return RPE_EUNSUPPORTED;
    const int P = Ho * Wo;
    const int cpb = channel_split(C, P, B);
    dim3 grid((P + 255) / 256, (C + cpb - 1) / cpb, B);
    if (src_is_u8)
        hipLaunchKernelGGL(resize_frames_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char *)src, B, C, H, W, Ho,
                           Wo, divisor, pair_split, cpb, 1.f, 1.f, 0, out);
    else
        hipLaunchKernelGGL(resize_frames_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, B, C, H, W, Ho, Wo, divisor,
                           pair_split, cpb, 1.f, 1.f, 0, out);
    return rpe_launch_status();
}

RPE_API int rpe_resize_flow2d(const float *flow, int B, int H, int W, int Ho, int Wo, float scale_x, float scale_y, float *out,
                              rpe_stream_t stream) {
    if (!flow || !out || B < 0 || H < 1 || W < 1 || Ho < 1 || Wo < 1) return RPE_EINVAL;
    if (B == 0) return 0;
    if (B > 65535 || (int64_t)Ho * Wo >= (1ll << 31) || (int64_t)H * W >= (1ll << 31)) return RPE_EUNSUPPORTED;
    dim3 grid((Ho * Wo + 255) / 256, 2, B);
    hipLaunchKernelGGL(resize_frames_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, flow, B, 2, H, W, Ho, Wo, 0.f, 0, 1, scale_x,
                       scale_y, 1, out);
    return rpe_launch_status();
}

RPE_API int rpe_upsample2x_pair(const float *a, int Ca, float scale_a, const float *b, int Cb, int B, int h, int w, float *out_a,
                                float *out_b, rpe_stream_t stream) {
    if (B < 0 || Ca < 0 || Cb < 0 || h < 1 || w < 1 || (Ca > 0 && (!a || !out_a)) || (Cb > 0 && (!b || !out_b))) return RPE_EINVAL;
    if (B == 0 || Ca + Cb == 0) return 0;
    if (B > 65535 || (int64_t)h * w * 4 >= (1ll << 31)) return RPE_EUNSUPPORTED;
    const int P = 4 * h * w;
    const int cpb = channel_split(Ca + Cb, P, B);
    dim3 grid((P + 255) / 256, (Ca + Cb + cpb - 1) / cpb, B);
    hipLaunchKernelGGL(upsample2x_pair_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, Ca, scale_a, b, Cb, h, w, cpb, out_a, out_b);
    return rpe_launch_status();
}

RPE_API int rpe_project_feat_nn_corr(const float *xy, int64_t xy_sb, int64_t xy_sd, int64_t xy_sn, const float *feat_2d, int C2,
                                           int H, int W, const float *sampled_2d, int64_t sm_sb, int64_t sm_sc, int64_t sm_sn,
                                           const float *feat_3d, int64_t f3_sb, int64_t f3_sc, int64_t f3_sn, int C3a,
                                           const float *feat_3d_b, int64_t g3_sb, int64_t g3_sc, int64_t g3_sn, int C3b, float scale_even,
                                           float scale_odd, float div_even, float div_odd, const int64_t *nn_idx, const float *subtract, int n_subtract, const float *append,
                                           int n_append, int B, int N, float *workspace, float *out, rpe_stream_t stream) {
    const int C3 = C3a + C3b;
    if (!xy || !feat_2d || !nn_idx || !out || !workspace || B < 0 || C2 < 1 || C3a < 0 || C3b < 0 || H < 1 || W < 1 || N < 1 ||
        (C3a > 0 && !feat_3d) || (C3b > 0 && !feat_3d_b))
        return RPE_EINVAL;
    const Feat3 F3{feat_3d, f3_sb, f3_sc, f3_sn, C3a, feat_3d_b, g3_sb, g3_sc, g3_sn, scale_even, scale_odd, div_even, div_odd};
    if (n_subtract < 0 || n_subtract > C3 || n_append < 0 || (n_subtract > 0 && !subtract) || (n_append > 0 && !append)) return RPE_EINVAL;
    if (reinterpret_cast<uintptr_t>(workspace) & 15) return RPE_EINVAL;
    if (B == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int RP = (((C2 + 3) & ~3) + ((C3 + 3) & ~3)) | 1;  // LDS pitch of point_rows_copy_kernel
    const int PB = 32;  // points per workgroup: 512 workgroups at level 1, 10.0 us (64 points: 10.9; one thread per point and slice: 14.5)
    if (sampled_2d && PB * RP * 4 <= 64 * 1024) {
        hipLaunchKernelGGL(point_rows_copy_kernel, dim3((N + PB - 1) / PB, B), dim3(256), (size_t)PB * RP * sizeof(float), st, sampled_2d, sm_sb,
                           sm_sc, sm_sn, C2, F3, C3, N, PB, workspace);
    } else {
        const int cpb = 8;  // channels per thread: 32-byte row segments, (C2+C3)/8 times the threads
        hipLaunchKernelGGL(point_rows_kernel, dim3(xcd_grid((C2 + C3 + cpb - 1) / cpb, (N + 255) / 256), 1, B), dim3(256), 0, st, xy, xy_sb,
                           xy_sd, xy_sn, feat_2d, C2, H, W, F3, C3, N, cpb, sampled_2d, sm_sb, sm_sc, sm_sn, workspace);
    }
    if ((int64_t)B * H * W <= 16384) {  // the coarse levels (up to 36 x 60 at batch 4): channels split over eight waves
        hipLaunchKernelGGL(project_rows_small_kernel<8>, dim3((H * W + 63) / 64, B), dim3(8 * 64), 0, st, xy, xy_sb, xy_sd, xy_sn, feat_2d, C2,
                           H, W, C3, N, workspace, nn_idx, subtract, n_subtract, append, n_append, out);
        return rpe_launch_status();
    }
    hipLaunchKernelGGL(project_rows_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, st, xy, xy_sb, xy_sd, xy_sn, feat_2d, C2,
                       H, W, C3, N, workspace, nn_idx, subtract, n_subtract, append, n_append, out);
    return rpe_launch_status();
}
