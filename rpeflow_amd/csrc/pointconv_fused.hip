// PointConv as ONE kernel for gfx950: neighbour gather + weight net + the 16 x (C+3) weighted sums + nn.Linear +
// bias / eval-BatchNorm / activation, for PointConvDownSampling.forward and PointConvNoSampling.forward
// (models/pointconv.py:33-61, 90-122).  The reference's [B,Q,16,C+3] gather and [B,Q,16(C+3)] product (208 MB each at
// pyramid level 1) never exist in HBM: a workgroup owns BM queries and walks the C+3 input channels 16 at a time.
//
//   rows  [B,M,CFp]  cat([xyz, features]) channel-last (pointconv.py:43-44), zero-padded to CFp = 16*ceil((C+3)/16)
//                    (rpe_pointconv_pack_rows, or the previous layer's rows-mode output)
//   stage 1, per query q and channel chunk:   G_q[w][c] = sum_j wn_q[j][w] * rows[knn[q][j]][c]          (:54-57)
//       one v_mfma_f32_16x16x4_f32 group per query: A = wn_q^T (16 w x 16 j), B = the 16 gathered row pieces
//       (16 j x 16 c, loaded straight into the operand layout: lane (j%4-slot, c)), D = G_q -> LDS tile A[q][16c + w]
//   stage 2, per chunk:   out[q][o] += sum_{c,w} A[q][16c + w] * L[o][w*(C+3) + c]                       (:58)
//       a plain MFMA GEMM step, M = BM queries, N = outputs, K = 256: A fragments are ds_read_b128 from the tile
//       (XOR-swizzled 16-byte blocks: conflict-free for the stage-1 writes and the stage-2 reads), B fragments are
//       1-KiB coalesced global loads from the linear weight pre-packed in fragment order (Lp, built once per module)
//   epilogue: y = act(scale[o]*out + shift[o]) (bias + eval BatchNorm folded by the caller), stored channel-first
//       [B,Cout,Q] or as the NEXT layer's rows [B,Q,stride] = [xyz_q | y | 0...].
//
// fp32 throughout (v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain); the k-order of the sums differs from the
// reference's matmul + Linear, so results agree to fp32 re-association (tests: 1e-4 relative), like the two-kernel path.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PcArgs {
    const float *rows;
    int CFp, nchunks, M;
    const int64_t *knn;
    int64_t knn_sq;
    const float *q_xyz;
    int64_t q_sb, q_sd, q_sn;
    const float *w1, *b1, *w2, *b2;  // weight net MLP2d(3,[8,16]): [8,3] [8] [16,8] [16]
    float slope;
    const f32x4 *Lp;  // [nchunks][16][NTT][64] float4
    int NTT;
    const float *scale, *shift;  // [Cout] or NULL
    int act;                     // 0 none, 1 relu, 2 leaky_relu(act_slope)
    float act_slope;
    float *out;
    int out_mode, out_stride;  // 0: [B,Cout,Q];  1: rows [B,Q,out_stride] = [xyz_q | y | zeros]
    int Cout, Q;
    int tiles, nsplit;  // query tiles per cloud, output splits: the grid is 1-D, tiles * nsplit * B workgroups (see the kernel)
};

__device__ __forceinline__ float leaky(float x, float slope) { return x >= 0.f ? x : x * slope; }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// KS groups of 4 waves: in a group WM x WN waves over (queries, outputs), a wave owns MT x NT tiles of 16 x 16.  KS = 2
// (the smallest levels, where a workgroup per 16 queries still leaves half the SIMDs idle): the two groups take alternate
// channel chunks with a tile each, and their partial sums meet in LDS at the end -- twice the waves on the same K loop.
template <int MT, int WM, int NT, int WN, int KS>
__global__ __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) void pointconv_fused_kernel(PcArgs a) {
    static_assert(WM * WN == 4, "four waves a group");
    constexpr int BM = 16 * MT * WM, QW = BM / 4;
    constexpr int PD = MT * NT >= 8 ? 2 : MT * NT >= 4 ? 4 : MT * NT >= 2 ? 8 : 16;  // B-fragment ring (divides the 16 groups of a chunk): PD - 1 groups in flight
    extern __shared__ __align__(16) float smem[];
    const int lane = rpe_lane();
    const int kg = rpe_uniform((int)(threadIdx.x >> 8));         // wave group
    const int wave = rpe_uniform((int)(threadIdx.x >> 6)) & 3;  // wave inside its group
    float *atile = smem + kg * (BM * 256);        // [BM][256] floats per group, 16-byte blocks XOR-swizzled by (row & 15)
    int *rowoff = (int *)(smem + KS * BM * 256);  // [BM][16] element offsets of the gathered rows (every group writes the same)
    const int kk = lane >> 4, n16 = lane & 15;
    // Workgroup ids go round the 8 XCDs (id % 8 shares an L2); remapped so that an XCD works through a CONTIGUOUS stretch of
    // (cloud, query tile, output split): its L2 then holds one cloud's rows instead of all of them, and the splits of a
    // query tile, which gather the same rows, run side by side.  (Pure placement: any id -> XCD assignment is correct.)
    const int nwg = (int)gridDim.x, orig = (int)blockIdx.x, xcd = orig & 7, per = nwg >> 3, rem = nwg & 7;
    const int wgid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (orig >> 3);
    const int bz = wgid % a.nsplit, bx = (wgid / a.nsplit) % a.tiles;
    const int b = wgid / (a.nsplit * a.tiles), q0 = bx * BM;
    const float *rows_b = a.rows + (int64_t)b * a.M * a.CFp;

    // ---- prologue: weight net of this wave's QW queries, in the stage-1 A-operand layout (lane = (j-slot kk, w = n16))
    float w2r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w2r[i] = a.w2[n16 * 8 + i];
    const float b2r = a.b2[n16];
    // Queries in batches of QB: all neighbour indices of a batch are requested first, then all neighbour coordinates, then
    // the arithmetic -- two memory round trips a batch instead of two per query.  The batch loop stays rolled and the
    // values are parked in this wave's own rows of the LDS tile (unrolled over 16 queries the compiler keeps hundreds
    // of uniform temporaries alive and spills).
    constexpr int QB = QW <= 8 ? QW : 4;  // (16 queries a wave: batches of 4, or the transient registers spill)
    float bv[QW][4];  // stage-1 B operands of the chunk in flight (first chunk: requested here, with the coordinates)
    const int first_chunk = 16 * min(kg, a.nchunks - 1);
#pragma unroll
    for (int ib = 0; ib < QW; ib += QB) {
        int off[QB][4];
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int q = min(q0 + wave * QW + ib + u, a.Q - 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) off[u][s] = (int)a.knn[((int64_t)b * a.Q + q) * a.knn_sq + 4 * s + kk] * a.CFp;
        }
#pragma unroll
        for (int u = 0; u < QB; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) bv[ib + u][s] = rows_b[off[u][s] + first_chunk + n16];
        float nx[QB][4][3];
#pragma unroll
        for (int u = 0; u < QB; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float *r = rows_b + off[u][s];
                nx[u][s][0] = r[0], nx[u][s][1] = r[1], nx[u][s][2] = r[2];
            }
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int ql = wave * QW + ib + u;
            const int q = min(q0 + ql, a.Q - 1);
            const float *qp = a.q_xyz + (int64_t)b * a.q_sb + (int64_t)q * a.q_sn;
            const float qx = qp[0], qy = qp[a.q_sd], qz = qp[2 * a.q_sd];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (n16 == 0) rowoff[ql * 16 + 4 * s + kk] = off[u][s];
                const float rel[3] = {nx[u][s][0] - qx, nx[u][s][1] - qy, nx[u][s][2] - qz};
                float h[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    float v = a.b1[o];
#pragma unroll
                    for (int d = 0; d < 3; ++d) v = __fmaf_rn(a.w1[o * 3 + d], rel[d], v);
                    h[o] = leaky(v, a.slope);
                }
                float v = b2r;
#pragma unroll
                for (int i = 0; i < 8; ++i) v = __fmaf_rn(w2r[i], h[i], v);
                atile[ql * 256 + s * 64 + lane] = leaky(v, a.slope);
            }
        }
    }
    __syncthreads();
    float wnA[QW][4];
#pragma unroll
    for (int iq = 0; iq < QW; ++iq)
#pragma unroll
        for (int s = 0; s < 4; ++s) wnA[iq][s] = atile[(wave * QW + iq) * 256 + s * 64 + lane];
    __syncthreads();  // every wave holds its values before stage 1 overwrites the tile

    const int wm = wave / WN, wn = wave % WN;
    const int tbase = bz * (WN * NT) + wn * NT;
    // Two-level sums: an MFMA accumulator is one sequential fp32 fma chain, so each chunk (K = 256) gets a fresh one and
    // the chunk totals are added up separately -- the blocked summation of a library GEMM, not a 3000-term chain.
    f32x4 total[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) total[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline over the channel chunks: the gathered row pieces of chunk ci + 1 are requested before stage 2 of
    // chunk ci and land under its MFMAs; stage 2 keeps PD groups of B fragments in flight (the smaller the wave tile, the
    // fewer MFMAs a group has to hide an L2 round trip, the deeper the ring).
    for (int c0 = 0; c0 < a.nchunks; c0 += KS) {
        const int ci = c0 + kg;
        const bool live = KS == 1 || ci < a.nchunks;  // (group-uniform; every wave still meets both barriers)
        // ---- stage 1: G for this chunk's 16 channels -> LDS tile
#pragma unroll
        for (int u = 0; u < QW; ++u) {
            f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) g = mfma16(wnA[u][s], bv[u][s], g);
            const int ql = wave * QW + u;
            const int blk = (4 * n16 + kk) ^ (ql & 15);
            *(f32x4 *)(atile + ql * 256 + 4 * blk) = g;
        }
        // Request order matters: vmcnt counts in order, and hipcc waits vmcnt(0) for the loop-carried gather registers at
        // the head of stage 1.  So (1) the first PD - 1 fragment groups of THIS chunk, whose round trip then runs under
        // the barrier, (2) the gather of the NEXT chunk, which has all of stage 2 to land; the fragments requested during
        // stage 2 are younger than both.  (In-kernel stamps: with the fragments requested in front of stage 1 that
        // vmcnt(0) sat on them for ~2000 cycles a chunk.)
        f32x4 bf[PD][NT];
        if (live) {
#pragma unroll
            for (int d = 0; d < PD - 1; ++d)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[d][nt] = a.Lp[(((int64_t)ci * 16 + d) * a.NTT + tbase + nt) * 64 + lane];
        }
        if (ci + KS < a.nchunks) {
#pragma unroll
            for (int u = 0; u < QW; ++u)
#pragma unroll
                for (int s = 0; s < 4; ++s) bv[u][s] = rows_b[rowoff[(wave * QW + u) * 16 + 4 * s + kk] + 16 * (ci + KS) + n16];
        }
        __syncthreads();
        if (!live) {  // past the last chunk: nothing to add, but the partner group still needs this round's second barrier
            __syncthreads();
            continue;
        }

        // ---- stage 2: total += A[BM x 256] * Lp chunk
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 acc_odd = f32x4{0.f, 0.f, 0.f, 0.f};  // MT*NT == 1: a second chain hides the 40-cycle dependent latency
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g + PD - 1 < 16) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[(g + PD - 1) % PD][nt] = a.Lp[(((int64_t)ci * 16 + g + PD - 1) * a.NTT + tbase + nt) * 64 + lane];
            }
            f32x4 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = 16 * (wm * MT + mt) + n16;
                af[mt] = *(const f32x4 *)(atile + row * 256 + 4 * ((4 * g + kk) ^ n16));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (MT * NT == 1 && (s & 1))
                            acc_odd = mfma16(af[mt][s], bf[g % PD][nt][s], acc_odd);
                        else
                            acc[mt][nt] = mfma16(af[mt][s], bf[g % PD][nt][s], acc[mt][nt]);
                    }
        }
        if (MT * NT == 1) acc[0][0] += acc_odd;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) total[mt][nt] += acc[mt][nt];
        __syncthreads();
    }

    if constexpr (KS == 2) {  // group 1 hands its partial sums over through LDS (the tiles are dead: the last round's barrier is behind)
        float *xchg = smem;
        if (kg == 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) *(f32x4 *)(xchg + ((wave * MT * NT + mt * NT + nt) * 64 + lane) * 4) = total[mt][nt];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) total[mt][nt] += *(const f32x4 *)(xchg + ((wave * MT * NT + mt * NT + nt) * 64 + lane) * 4);
    }

    // ---- epilogue: lane holds out[q = tile row 4*kk + r][o = 16*t + n16]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int o = 16 * (tbase + nt) + n16;
        const bool o_live = o < a.Cout;
        const float sc = (o_live && a.scale) ? a.scale[o] : 1.f, sh = (o_live && a.shift) ? a.shift[o] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int qr = q0 + 16 * (wm * MT + mt) + 4 * kk;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = total[mt][nt][r] * sc + sh;
                if (a.act == 1) v = rpe_relu(v);
                if (a.act == 2) v = leaky(v, a.act_slope);
                y[r] = v;
            }
            if (a.out_mode == 0) {
                if (!o_live) continue;
                float *dst = a.out + ((int64_t)b * a.Cout + o) * a.Q + qr;
                if (qr + 3 < a.Q && (a.Q & 3) == 0) {
                    *(f32x4 *)dst = f32x4{y[0], y[1], y[2], y[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (qr + r < a.Q) dst[r] = y[r];
                }
            } else {
                if (3 + o >= a.out_stride) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (qr + r < a.Q) a.out[((int64_t)b * a.Q + qr + r) * a.out_stride + 3 + o] = o_live ? y[r] : 0.f;
            }
        }
    }
    if (a.out_mode == 1 && bz == 0) {
        // the rows' xyz columns and the zero tail beyond the last output tile
        const int covered = 3 + 16 * a.nsplit * WN * NT;
        for (int i = threadIdx.x; i < BM * 16; i += 256) {  // (KS == 2: group 0 only is left, threadIdx.x < 256)
            const int ql = i >> 4, c = i & 15, q = q0 + ql;
            if (q >= a.Q) continue;
            float *row = a.out + ((int64_t)b * a.Q + q) * a.out_stride;
            if (c < 3) row[c] = a.q_xyz[(int64_t)b * a.q_sb + c * a.q_sd + (int64_t)q * a.q_sn];
            for (int z = covered + c; z < a.out_stride; z += 16) row[z] = 0.f;
        }
    }
}

// rows[b][m][:] = [xyz[b][:,m] | src0[b][:,m] | src1 ... | 0 ...]: the concatenation + transposition of pointconv.py:43-44
// (and of the callers' torch.cat in front of it, RPEFlow_core.py:382-391) in one pass; a 64-point x 16-channel tile is
// read along the points and written along the channels through LDS.
struct PackSrc {
    const float *p[5];
    int64_t sb[5], sc[5], sn[5];
    int c0[6];  // first row column of source i; c0[n_src] = total
    int n_src;
};

__global__ __launch_bounds__(256) void pointconv_pack_kernel(PackSrc s, int M, int CFp, float *__restrict__ rows) {
    __shared__ float tile[16][65];
    const int b = blockIdx.z, m0 = blockIdx.x * 64, cbase = blockIdx.y * 16;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // read: tx = point, ty = channel quad
    const int total = s.c0[s.n_src];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = cbase + ty * 4 + i, m = m0 + tx;
        float v = 0.f;
        if (c < total && m < M) {
            int k = 0;
#pragma unroll
            for (int t = 1; t < 5; ++t)
                if (t < s.n_src && c >= s.c0[t]) k = t;
            v = s.p[k][(int64_t)b * s.sb[k] + (int64_t)(c - s.c0[k]) * s.sc[k] + (int64_t)m * s.sn[k]];
        }
        tile[ty * 4 + i][tx] = v;
    }
    __syncthreads();
    const int wc = threadIdx.x & 15, wm = threadIdx.x >> 4;  // write: wc = channel, wm = point (16 per pass)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm + 16 * i;
        if (m < M) rows[((int64_t)b * M + m) * CFp + cbase + wc] = tile[wc][wm + 16 * i];
    }
}

template <int MT, int WM, int NT, int WN, int KS = 1>
int launch_fused(PcArgs a, int B, int nsplit, hipStream_t st) {
    constexpr int BM = 16 * MT * WM;
    constexpr int smem = KS * BM * 256 * 4 + BM * 16 * 4;
    // per launch: the attribute belongs to the CURRENT device (a process may drive several GPUs), as knn / fps / correlation set theirs
    hipError_t e = hipFuncSetAttribute((const void *)pointconv_fused_kernel<MT, WM, NT, WN, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    a.tiles = (a.Q + BM - 1) / BM;
    a.nsplit = nsplit;
    const int64_t nwg = (int64_t)a.tiles * nsplit * B;
    if (nwg >= (1ll << 31)) return RPE_EUNSUPPORTED;
    hipLaunchKernelGGL((pointconv_fused_kernel<MT, WM, NT, WN, KS>), dim3((unsigned)nwg), dim3(256 * KS), smem, st, a);
    return rpe_launch_status();
}

}  // namespace

RPE_API int rpe_pointconv_pack_rows(const float *xyz, int64_t x_sb, int64_t x_sd, int64_t x_sn, const float *const *srcs,
                                    const int64_t *src_strides, const int *src_channels, int n_src, int B, int M, int CFp,
                                    float *rows, rpe_stream_t stream) {
    if (!xyz || !rows || n_src < 0 || n_src > 4 || (n_src && (!srcs || !src_strides || !src_channels))) return RPE_EINVAL;
    if (B < 0 || M < 1 || CFp < 16 || (CFp & 15)) return RPE_EINVAL;
    PackSrc s;
    s.p[0] = xyz, s.sb[0] = x_sb, s.sc[0] = x_sd, s.sn[0] = x_sn, s.c0[0] = 0;
    int total = 3;
    for (int i = 0; i < n_src; ++i) {
        if (!srcs[i] || src_channels[i] < 1) return RPE_EINVAL;
        s.p[i + 1] = srcs[i], s.sb[i + 1] = src_strides[3 * i], s.sc[i + 1] = src_strides[3 * i + 1], s.sn[i + 1] = src_strides[3 * i + 2];
        s.c0[i + 1] = total;
        total += src_channels[i];
    }
    for (int i = n_src + 1; i < 5; ++i) s.p[i] = xyz, s.sb[i] = s.sc[i] = s.sn[i] = 0, s.c0[i] = total;
    s.c0[n_src + 1] = total;
    s.c0[5] = total;
    s.n_src = n_src + 1;
    if (total > CFp) return RPE_EINVAL;
    if (B == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    hipLaunchKernelGGL(pointconv_pack_kernel, dim3((M + 63) / 64, CFp / 16, B), dim3(256), 0, (hipStream_t)stream, s, M, CFp, rows);
    return rpe_launch_status();
}

RPE_API int rpe_pointconv_fused(const float *rows, int CFp, int M, const int64_t *knn, int64_t knn_row_stride, const float *q_xyz,
                                int64_t q_sb, int64_t q_sd, int64_t q_sn, const float *w1, const float *b1, const float *w2,
                                const float *b2, float leaky_slope, const float *packed_linear, int n_tiles, const float *scale,
                                const float *shift, int act, float act_slope, int B, int Q, int Cout, int out_mode, int out_stride,
                                float *out, rpe_stream_t stream) {
    if (!rows || !knn || !q_xyz || !w1 || !b1 || !w2 || !b2 || !packed_linear || !out) return RPE_EINVAL;
    if (B < 0 || Q < 0 || M < 1 || Cout < 1 || CFp < 16 || (CFp & 15) || knn_row_stride < 16 || act < 0 || act > 2) return RPE_EINVAL;
    if (n_tiles < 8 || (n_tiles & 7) || 16 * n_tiles < Cout) return RPE_EINVAL;
    if (out_mode != 0 && (out_mode != 1 || out_stride < 3 + Cout)) return RPE_EINVAL;
    if ((int64_t)M * CFp >= (1ll << 31)) return RPE_EUNSUPPORTED;
    if (B == 0 || Q == 0) return 0;
    if (B > 65535) return RPE_EUNSUPPORTED;
    PcArgs a{rows, CFp, CFp / 16, M, knn, knn_row_stride, q_xyz, q_sb, q_sd, q_sn, w1, b1, w2, b2, leaky_slope,
             (const f32x4 *)packed_linear, n_tiles, scale, shift, act, act_slope, out, out_mode, out_stride, Cout, Q, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    // Tile choice: BN = 128 outputs per workgroup (64 for narrow layers) and the largest BM in {64, 32, 16} queries that
    // still gives every CU two workgroups (one in its gather / stage 1 while the other multiplies); the smallest levels
    // split the outputs over more workgroups instead (each repeats stage 1, which is 16/BN of the work).
    const int64_t work = (int64_t)B * Q;
    const int n128 = (Cout + 127) / 128, n64 = (Cout + 63) / 64;
    const int64_t want = 2 * 256;
    if (Cout <= 64) {
        if (work >= 64 * want) return launch_fused<2, 2, 2, 2>(a, B, 1, st);
        if (work >= 32 * want) return launch_fused<1, 2, 2, 2>(a, B, 1, st);
        if (work >= 16 * want) return launch_fused<1, 1, 1, 4>(a, B, 1, st);
        return launch_fused<1, 1, 1, 4, 2>(a, B, 1, st);
    }
    if (work * n128 >= 64 * want) return launch_fused<2, 2, 4, 2>(a, B, n128, st);
    if (work * n128 >= 32 * want) return launch_fused<1, 2, 4, 2>(a, B, n128, st);
    if (work * n128 >= 16 * want) return launch_fused<1, 1, 2, 4>(a, B, n128, st);
    if (work * n64 >= 16 * want) return launch_fused<1, 1, 1, 4>(a, B, n64, st);
    return launch_fused<1, 1, 1, 4, 2>(a, B, n64, st);  // fewer than two workgroups a CU even so: split the channel chunks over two wave groups
}
