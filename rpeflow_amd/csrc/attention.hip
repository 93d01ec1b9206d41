// Channel ("transposed") attention core of the Restormer cross blocks: Mutual_Attention{2D,3D}.forward,
// models/restormer_arch.py:169-204 and 251-283.
//
// Reference chain per block:  q,k = F.normalize(q,k over the P positions);  attn = softmax(q k^T * temperature);
// out = project_out(attn v);  ~12 PyTorch kernels over [B,C,P] tensors.  Here:
//
//   attn_gram_kernel     G[b][h] = q_h k_h^T (c x c, contraction over P) and the squared row norms of q and k, in one
//                        pass over q and k on v_mfma_f32_16x16x4_f32.  Normalising afterwards,
//                        G_ij / (max(|q_i|,eps) max(|k_j|,eps)), equals the reference's normalise-then-multiply up to
//                        fp32 rounding.  Positions are split over workgroups; partial sums are written per workgroup
//                        and added in a fixed order by the next kernel (deterministic, no atomics).
//   attn_softmax_kernel  sums the partials, scales by the norms and the temperature, softmax over j (one wave per row).
//   attn_project_kernel  folds the 1x1 project_out convolution in: M[b] = W_o blockdiag_h(attn_h)  (C x C), so that the
//                        caller finishes the block with ONE batched GEMM  out = residual + M[b] v[b].
#include <math.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// positions per workgroup (4 waves x 16-position groups, interleaved).  On the coarse maps a launch is a handful of workgroups
// and a wave's loop over its positions is a chain of load-wait-multiply rounds: shorter chunks there (more workgroups, 4 rounds
// a wave instead of up to 32; the partial grams are added in chunk order by the softmax kernel either way).
__host__ __device__ inline int attn_chunk(int64_t P) { return P <= 16384 ? 256 : 2048; }
__host__ __device__ inline int attn_chunks(int64_t P) { return (int)((P + attn_chunk(P) - 1) / attn_chunk(P)); }

template <bool VEC>
__device__ __forceinline__ float4 load_row4(const float *__restrict__ base, int row, int c, int64_t P, int64_t p, int64_t pend) {
    // 4 consecutive positions p..p+3 of channel `row`; zero outside the head's channels / the chunk
    if (row >= c || p >= pend) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float *src = base + (int64_t)row * P + p;
    if (VEC) return *reinterpret_cast<const float4 *>(src);  // P % 4 == 0, 16-byte aligned planes: p + 3 < pend
    float4 v;
    v.x = src[0];
    v.y = p + 1 < pend ? src[1] : 0.f;
    v.z = p + 2 < pend ? src[2] : 0.f;
    v.w = p + 3 < pend ? src[3] : 0.f;
    return v;
}

// grid (chunks * ceil(T / R), heads, B), 256 threads; T = ceil(c / 16).  Workgroup (s, tg) owns R tile rows (16 R rows) of the
// gram for positions [s * chunk, (s + 1) * chunk): R q tile rows against all T k tiles (k is re-read ceil(T / R) times, from L2).
template <int T, int R, bool VEC>
__global__ __launch_bounds__(256) void attn_gram_kernel(const float *__restrict__ q, const float *__restrict__ k, int64_t batch_stride,
                                                        int heads, int c, int64_t P, float *__restrict__ gpart,
                                                        float *__restrict__ npart) {
    constexpr int TG = (T + R - 1) / R;  // workgroups per chunk: each owns R of the T tile rows (k is read TG times)
    __shared__ float tiles[R * T * 256];  // [R][T][64][4] accumulator tiles
    __shared__ float norms[(T + R) * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x / TG, tg = blockIdx.x - s * TG, h = blockIdx.y, b = blockIdx.z, S = gridDim.x / TG;
    const int ti0 = tg * R;
    const int row = lane & 15, grp = lane >> 4;
    const float *qh = q + (int64_t)b * batch_stride + (int64_t)h * c * P;
    const float *kh = k + (int64_t)b * batch_stride + (int64_t)h * c * P;
    const int chunk = attn_chunk(P);
    const int64_t p0 = (int64_t)s * chunk;
    const int64_t pend = p0 + chunk < P ? p0 + chunk : P;
    const bool k_norms = tg == 0;  // the k norms are the same in every group of tile rows: the first one writes them

    f32x4 acc[R][T];
    float sq[R], sk[T];
#pragma unroll
    for (int j = 0; j < T; ++j) sk[j] = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        sq[r] = 0.f;
#pragma unroll
        for (int j = 0; j < T; ++j) acc[r][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto accumulate = [&](const float4 (&a)[R], const float4 (&bb)[T]) {
#pragma unroll
        for (int r = 0; r < R; ++r) sq[r] += (a[r].x * a[r].x + a[r].y * a[r].y) + (a[r].z * a[r].z + a[r].w * a[r].w);
        if (k_norms) {
#pragma unroll
            for (int t = 0; t < T; ++t) sk[t] += (bb[t].x * bb[t].x + bb[t].y * bb[t].y) + (bb[t].z * bb[t].z + bb[t].w * bb[t].w);
        }
        // position phase outermost: consecutive MFMAs write different accumulators (40-cycle dependent latency)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < T; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].x, bb[j].x, acc[r][j], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < T; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].y, bb[j].y, acc[r][j], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < T; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].z, bb[j].z, acc[r][j], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < T; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].w, bb[j].w, acc[r][j], 0, 0, 0);
    };
    int64_t gs = p0 + wave * 16;  // wave-uniform trip counts below: every lane joins the MFMAs
    if (VEC) {
        // Full 16-position groups, no guards: a row beyond the head's channels reads the last valid row instead (its gram
        // rows / columns and norms are never stored); the next group's loads are issued before this group's MFMAs.
        const float *qrow[R];
#pragma unroll
        for (int r = 0; r < R; ++r) qrow[r] = qh + (int64_t)min((ti0 + r) * 16 + row, c - 1) * P + grp * 4;
        const float *krow[T];
#pragma unroll
        for (int t = 0; t < T; ++t) krow[t] = kh + (int64_t)min(t * 16 + row, c - 1) * P + grp * 4;
        if (gs + 16 <= pend) {
            float4 a[R], bb[T];
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = *reinterpret_cast<const float4 *>(qrow[r] + gs);
#pragma unroll
            for (int t = 0; t < T; ++t) bb[t] = *reinterpret_cast<const float4 *>(krow[t] + gs);
            for (; gs + 64 + 16 <= pend; gs += 64) {
                float4 na[R], nb[T];
#pragma unroll
                for (int r = 0; r < R; ++r) na[r] = *reinterpret_cast<const float4 *>(qrow[r] + gs + 64);
#pragma unroll
                for (int t = 0; t < T; ++t) nb[t] = *reinterpret_cast<const float4 *>(krow[t] + gs + 64);
                accumulate(a, bb);
#pragma unroll
                for (int r = 0; r < R; ++r) a[r] = na[r];
#pragma unroll
                for (int t = 0; t < T; ++t) bb[t] = nb[t];
            }
            accumulate(a, bb);
            gs += 64;
        }
    }
    for (; gs < pend; gs += 64) {  // the chunk's ragged end (and everything when rows are not 16-byte aligned)
        const int64_t p = gs + grp * 4;
        float4 a[R], bb[T];
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = load_row4<VEC>(qh, (ti0 + r) * 16 + row, c, P, p, pend);
#pragma unroll
        for (int t = 0; t < T; ++t) bb[t] = load_row4<VEC>(kh, t * 16 + row, c, P, p, pend);
        accumulate(a, bb);
    }
    // norms: add the four position groups of the wave (lanes l, l^16, l^32, l^48 share a row)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        sq[r] += __shfl_xor(sq[r], 16);
        sq[r] += __shfl_xor(sq[r], 32);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        sk[t] += __shfl_xor(sk[t], 16);
        sk[t] += __shfl_xor(sk[t], 32);
    }
    for (int w = 0; w < 4; ++w) {  // waves add into LDS one after the other (fixed order)
        if (wave == w) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    f32x4 *slot = reinterpret_cast<f32x4 *>(tiles + ((r * T + j) * 64 + lane) * 4);
                    *slot = w == 0 ? acc[r][j] : *slot + acc[r][j];
                }
            if (grp == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) norms[r * 16 + row] = w == 0 ? sq[r] : norms[r * 16 + row] + sq[r];
#pragma unroll
                for (int t = 0; t < T; ++t) norms[(R + t) * 16 + row] = w == 0 ? sk[t] : norms[(R + t) * 16 + row] + sk[t];
            }
        }
        __syncthreads();
    }
    // partial gram [b][h][s][c][c]; tile element (j-tile, lane, reg): row 16 ti + 4 (lane >> 4) + reg, col 16 jt + (lane & 15)
    float *gp = gpart + (((int64_t)b * heads + h) * S + s) * c * c;
    const int rows = max(0, min(16 * R, c - ti0 * 16));  // rows of this group inside the head
    for (int o = threadIdx.x; o < rows * c; o += 256) {
        const int ig = o / c, j = o - ig * c, jt = j >> 4, jc = j & 15, r = ig >> 4, ir = ig & 15;
        gp[(ti0 * 16 + ig) * c + j] = tiles[((r * T + jt) * 64 + ((ir >> 2) << 4) + jc) * 4 + (ir & 3)];
    }
    // partial squared norms [b][s][2][heads*c]
    float *np = npart + ((int64_t)b * S + s) * 2 * heads * c + h * c;
    if ((int)threadIdx.x < rows) np[ti0 * 16 + threadIdx.x] = norms[threadIdx.x];
    if (k_norms)
        for (int i = threadIdx.x; i < c; i += 256) np[heads * c + i] = norms[R * 16 + i];
}

// One wave per attention row (b, h, i): sums the partials in a fixed order, scales by the norms and the temperature,
// softmax over j.  grid (ceil(B*C / 4)), 256 threads; attn_out [B][C][c].
__global__ __launch_bounds__(256) void attn_softmax_kernel(const float *__restrict__ gpart, const float *__restrict__ npart, int S,
                                                           const float *__restrict__ temperature, int B, int heads, int c, float eps,
                                                           float *__restrict__ attn_out) {
    const int lane = threadIdx.x & 63, C = heads * c;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * C) return;
    const int b = r / C, hi = r - b * C, h = hi / c, i = hi - h * c;
    const float *gp = gpart + ((int64_t)b * heads + h) * S * c * c + i * c;
    const float *np = npart + (int64_t)b * S * 2 * C;
    // the S partials are added in chunk order, their loads issued eight chunks at a time: one at a time the sums are
    // chains of S memory round trips (the partials were written on other XCDs: every load misses this one's L2)
    constexpr int TU = 8;
    float nq = 0.f;
    float g[2] = {0.f, 0.f}, nk[2] = {0.f, 0.f};
    const int j0 = min(lane, c - 1), j1 = min(lane + 64, c - 1);
    for (int t0 = 0; t0 < S; t0 += TU) {
        float a[TU], g0[TU], g1[TU], k0[TU], k1[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int64_t t = min(t0 + u, S - 1);
            a[u] = np[t * 2 * C + hi];
            g0[u] = gp[t * c * c + j0];
            k0[u] = np[t * 2 * C + C + h * c + j0];
            g1[u] = gp[t * c * c + j1];
            k1[u] = np[t * 2 * C + C + h * c + j1];
        }
#pragma unroll
        for (int u = 0; u < TU; ++u)
            if (t0 + u < S) {
                nq += a[u];
                g[0] += g0[u];
                nk[0] += k0[u];
                g[1] += g1[u];
                nk[1] += k1[u];
            }
    }
    nq = fmaxf(sqrtf(nq), eps);
    const float temp = temperature[h];
    float v[2];  // c <= 96 < 128: columns lane and lane + 64
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = lane + 64 * u;
        v[u] = j < c ? g[u] / (nq * fmaxf(sqrtf(nk[u]), eps)) * temp : -INFINITY;
        mx = fmaxf(mx, v[u]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        v[u] = lane + 64 * u < c ? expf(v[u] - mx) : 0.f;
        sum += v[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (lane + 64 * u < c) attn_out[(int64_t)r * c + lane + 64 * u] = v[u] * inv;
}

// M[b][o][h*c + j] = sum_i W_o[o][h*c + i] * attn[b][h][i][j].  grid (B, slices of M's elements), 256 threads;
// dynamic LDS: attn[b] (C*c floats).  PACKED: M[b] is written in rpe_pointwise_conv's weight-fragment order
// ([ceil(C/16)][ceil(C/4)][64]: entry (ot, kt, 16 k + i) = M[16 ot + i][4 kt + k], zero outside), so that the block's
// "x + M[b] v[b]" is ONE launch of the 1x1 kernel with the residual in its epilogue instead of a copy + a batched GEMM.
template <bool PACKED>
__global__ __launch_bounds__(256) void attn_project_kernel(const float *__restrict__ attn_in, const float *__restrict__ w_out, int heads,
                                                           int c, float *__restrict__ m_out) {
    extern __shared__ float attn[];
    const int C = heads * c, b = blockIdx.x;
    for (int o = threadIdx.x; o < C * c; o += 256) attn[o] = attn_in[(int64_t)b * C * c + o];
    __syncthreads();
    const int kt_n = (C + 3) / 4, total = PACKED ? ((C + 15) / 16) * kt_n * 64 : C * C;
    float *mb = m_out + (int64_t)b * total;
    const int per = (total + gridDim.y - 1) / gridDim.y;
    const int e_end = min(total, (int)(blockIdx.y + 1) * per);
    for (int e = blockIdx.y * per + threadIdx.x; e < e_end; e += 256) {
        int o, col;
        if (PACKED) {
            const int i = e & 15, k = (e >> 4) & 3, piece = e >> 6, kt = piece % kt_n, ot = piece / kt_n;
            o = 16 * ot + i, col = 4 * kt + k;
            if (o >= C || col >= C) {
                mb[e] = 0.f;
                continue;
            }
        } else {
            o = e / C, col = e - o * C;
        }
        const int h = col / c, j = col - h * c;
        const float *w = w_out + (int64_t)o * C + h * c;
        const float *a = attn + (h * c) * c + j;
        float s0 = 0.f, s1 = 0.f;
        int i = 0;
        for (; i + 1 < c; i += 2) {
            s0 = __fmaf_rn(w[i], a[i * c], s0);
            s1 = __fmaf_rn(w[i + 1], a[(i + 1) * c], s1);
        }
        if (i < c) s0 = __fmaf_rn(w[i], a[i * c], s0);
        mb[e] = s0 + s1;
    }
}

template <int T>
int launch_gram(bool vec, dim3 grid, hipStream_t st, const float *q, const float *k, int64_t batch_stride, int heads, int c, int64_t P,
                float *gpart, float *npart) {
    // big maps: a workgroup owns R tile rows of the gram, so that k is read ceil(T / R) times instead of T times (level 1,
    // with the softmax / project launches: c = 48: 49.5 -> 41.9 us; c = 81: 69.3 -> 59.1; c = 32: 35.4 -> 29.1; three rows of
    // six, or 1024-position chunks, were no better: 68 us)
    constexpr int R = T <= 3 ? T : 2;
    if (R > 1 && P > 16384) {
        grid.x *= (T + R - 1) / R;
        if (vec) hipLaunchKernelGGL((attn_gram_kernel<T, R, true>), grid, dim3(256), 0, st, q, k, batch_stride, heads, c, P, gpart, npart);
        else hipLaunchKernelGGL((attn_gram_kernel<T, R, false>), grid, dim3(256), 0, st, q, k, batch_stride, heads, c, P, gpart, npart);
        return rpe_launch_status();
    }
    grid.x *= T;
    if (vec) hipLaunchKernelGGL((attn_gram_kernel<T, 1, true>), grid, dim3(256), 0, st, q, k, batch_stride, heads, c, P, gpart, npart);
    else hipLaunchKernelGGL((attn_gram_kernel<T, 1, false>), grid, dim3(256), 0, st, q, k, batch_stride, heads, c, P, gpart, npart);
    return rpe_launch_status();
}

}  // namespace

RPE_API int64_t rpe_channel_attention_workspace_floats(int B, int heads, int c, int64_t P) {
    if (B < 0 || heads < 1 || c < 1 || P < 0) return -1;
    return (int64_t)B * attn_chunks(P > 0 ? P : 1) * ((int64_t)heads * c * c + 2 * (int64_t)heads * c) + (int64_t)B * heads * c * c;
}

namespace {
int attention_matrix(const float *q, const float *k, int64_t batch_stride, const float *temperature, const float *w_out, int B, int heads, int c,
                     int64_t P, float eps, float *workspace, float *m_out, bool packed, rpe_stream_t stream) {
    if (!q || !k || !temperature || !w_out || !workspace || !m_out || B < 0 || heads < 1 || c < 1 || P < 1) return RPE_EINVAL;
    if (B == 0) return 0;
    if (c > 96 || heads > 65535 || B > 65535) return RPE_EUNSUPPORTED;
    const int C = heads * c, S = attn_chunks(P);
    if ((size_t)C * c * sizeof(float) > 64 * 1024) return RPE_EUNSUPPORTED;  // attn[b] staged in LDS by attn_project_kernel
    float *gpart = workspace, *npart = workspace + (int64_t)B * heads * S * c * c;
    const bool vec = P % 4 == 0 && batch_stride % 4 == 0 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) == 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(S, heads, B);
    const int T = (c + 15) / 16;
    int rc;
    switch (T) {
        case 1: rc = launch_gram<1>(vec, grid, st, q, k, batch_stride, heads, c, P, gpart, npart); break;
        case 2: rc = launch_gram<2>(vec, grid, st, q, k, batch_stride, heads, c, P, gpart, npart); break;
        case 3: rc = launch_gram<3>(vec, grid, st, q, k, batch_stride, heads, c, P, gpart, npart); break;
        case 4: rc = launch_gram<4>(vec, grid, st, q, k, batch_stride, heads, c, P, gpart, npart); break;
        case 5: rc = launch_gram<5>(vec, grid, st, q, k, batch_stride, heads, c, P, gpart, npart); break;
        default: rc = launch_gram<6>(vec, grid, st, q, k, batch_stride, heads, c, P, gpart, npart); break;
    }
    if (rc) return rc;
    float *attn = npart + (int64_t)B * S * 2 * C;
    hipLaunchKernelGGL(attn_softmax_kernel, dim3((B * C + 3) / 4), dim3(256), 0, st, gpart, npart, S, temperature, B, heads, c, eps, attn);
    if (packed) hipLaunchKernelGGL(attn_project_kernel<true>, dim3(B, C >= 64 ? 16 : 4), dim3(256), (size_t)C * c * sizeof(float), st, attn, w_out, heads, c, m_out);
    else hipLaunchKernelGGL(attn_project_kernel<false>, dim3(B, C >= 64 ? 16 : 4), dim3(256), (size_t)C * c * sizeof(float), st, attn, w_out, heads, c, m_out);
    return rpe_launch_status();
}
}  // namespace

RPE_API int rpe_channel_attention_matrix(const float *q, const float *k, int64_t batch_stride, const float *temperature,
                                         const float *w_out, int B, int heads, int c, int64_t P, float eps, float *workspace,
                                         float *m_out, int packed, rpe_stream_t stream) {
    return attention_matrix(q, k, batch_stride, temperature, w_out, B, heads, c, P, eps, workspace, m_out, packed != 0, stream);
}
