// k_nearest_neighbor, k = 1, D = 2, on a cloud binned into a uniform cell grid (reached through rpe_knn's workspace): the model's nearest-projected-point search
// (RPEFlow_core.py:327-330 through wrapper.py:106-127: every pixel of a feature map asks for the nearest of N projected
// points; 34 560 x 4096 pairs per sample at pyramid level 1, twice per level).  The sweeping kernels of knn.hip evaluate
// every pair; here a wave's 64 queries only meet the points of the cells around them.
//
// Results are EXACTLY rpe_knn's (k = 1: the smallest distance as wrapper.py:49-51 rounds it, the lowest index among equal
// ones -- std::partial_sort with a one-element heap), for ANY queries: nothing below assumes a raster.  Raster-ordered
// queries are merely the fast case, because 64 consecutive queries then have a small bounding box.
//
//   build   one workgroup per cloud.  Domain = bounding box of a strided sample of the queries (any domain is valid: points
//           outside it fall into the border cells), about two points per cell, at most kMaxCells cells.  Counting sort by cell
//           (LDS histogram, scan, scatter): records (x, y, |p|^2, index) in cell-major, row-major order, so the points of a run
//           of cells of one cell row are ONE contiguous range.  Points with a non-finite coordinate are left out: their
//           distance is NaN or +inf for every query and `d < best` never takes them in the sweeping kernels either.
//   search  one wave per 16 consecutive queries, each query on four lanes that share out the candidates.  The wave's query box,
//           widened by one cell, is visited first: every lane evaluates its query against its share of the records with the
//           reference fallback's arithmetic (rpe_pair_dist) and the (distance, index) order.  Then the bound:
//           a point that could still win or tie for a lane has computed distance <= that lane's best, and the computed
//           distance differs from the true squared distance by at most 2^-19 (|q|^2 + |p|^2) (four times the worst case of the
//           five roundings, the margin of knn_grid.h); with |p|^2 <= 2 |q|^2 + 2 true that puts it within
//           R^2 = (worst best + 2^-17 max|q|^2) (1 + 2^-16) of ITS query, hence inside the query box widened by R.  The cell
//           coordinate is a monotone function of the point coordinate (two rounded operations, a clamp, a truncation), so the
//           cells of that widened box, computed with the same function, contain every such point -- also points and queries
//           outside the domain.  The cells not yet visited are visited; the bests only fall, so that is the end.
//           A lane without any candidate after the first visit (worst = +inf) widens the rectangle geometrically first.
#include <math.h>

#include "common.h"

namespace {

constexpr int kMaxCells = 4096;
constexpr int kBuildThreads = 1024;
constexpr int kHeaderBytes = 64;
constexpr int kPadRecords = 16;  // +inf records behind the last one: the search reads ranges in groups without a tail test
constexpr int kStartsBytes = ((kMaxCells + 2) * 4 + 15) / 16 * 16;

struct BinHeader {  // per cloud, written by the build kernel
    float x0, y0, inv_cw, inv_ch;
    int gx, gy, n_finite, pad;
};

__host__ __device__ inline int64_t ws_stride_bytes(int M) { return kHeaderBytes + kStartsBytes + 16ll * ((int64_t)M + kPadRecords); }

// cell coordinate of a point coordinate: monotone non-decreasing in v (fl(v - v0), fl(. * inv), clamp, truncation)
__device__ __forceinline__ int cell_coord(float v, float v0, float inv, int g) {
    float t = (v - v0) * inv;
    t = fminf(fmaxf(t, 0.f), (float)(g - 1));
    return (int)t;
}

__device__ __forceinline__ bool finite2(float x, float y) { return fabsf(x) <= 3.4028234e38f && fabsf(y) <= 3.4028234e38f; }

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

__global__ __launch_bounds__(kBuildThreads) void nearest2d_build_kernel(const float *__restrict__ inp, int64_t in_sb, int64_t in_sn, int64_t in_sd,
                                                                        const float *__restrict__ qry, int64_t q_sb, int64_t q_sn, int64_t q_sd,
                                                                        int M, int Q, char *__restrict__ ws, int64_t ws_stride) {
    __shared__ int cnt[kMaxCells + 1];
    __shared__ float red[4][kBuildThreads / RPE_WAVE];
    __shared__ int wsum[kBuildThreads / RPE_WAVE];
    __shared__ BinHeader hdr;
    const int tid = threadIdx.x, lane = rpe_lane(), wave = tid >> 6;
    const int b = blockIdx.x;
    inp += (int64_t)b * in_sb;
    qry += (int64_t)b * q_sb;
    char *base = ws + (int64_t)b * ws_stride;
    int *starts = (int *)(base + kHeaderBytes);
    float4 *recs = (float4 *)(base + kHeaderBytes + kStartsBytes);

    // 0. the first kHold points of every thread: requested now (beside the query sample), kept in registers for both passes
    constexpr int kHold = 8;
    float hx[kHold], hy[kHold];
#pragma unroll
    for (int j = 0; j < kHold; ++j) {
        const int i = tid + j * kBuildThreads;
        hx[j] = hy[j] = NAN;
        if (i < M) hx[j] = inp[(int64_t)i * in_sn], hy[j] = inp[(int64_t)i * in_sn + in_sd];
    }

    // 1. domain: bounding box of up to kBuildThreads queries, evenly strided, first and last included
    float mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    const int ns = Q < kBuildThreads ? Q : kBuildThreads;
    if (tid < ns) {
        const int step = ns > 1 ? (Q - 1) / (ns - 1) : 0;  // (32-bit, wave-uniform: no 64-bit division per thread)
        const int qi = tid == ns - 1 ? Q - 1 : tid * step;
        const float x = qry[(int64_t)qi * q_sn], y = qry[(int64_t)qi * q_sn + q_sd];
        if (finite2(x, y)) mnx = mxx = x, mny = mxy = y;
    }
    mnx = wave_min(mnx), mxx = wave_max(mxx), mny = wave_min(mny), mxy = wave_max(mxy);
    if (lane == 0) red[0][wave] = mnx, red[1][wave] = mxx, red[2][wave] = mny, red[3][wave] = mxy;
    for (int i = tid; i <= kMaxCells; i += kBuildThreads) cnt[i] = 0;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kBuildThreads / RPE_WAVE; ++w)
            mnx = fminf(mnx, red[0][w]), mxx = fmaxf(mxx, red[1][w]), mny = fminf(mny, red[2][w]), mxy = fmaxf(mxy, red[3][w]);
        if (!(mnx <= mxx)) mnx = 0.f, mxx = 1.f, mny = 0.f, mxy = 1.f;  // no finite query in the sample
        const float w = fmaxf(mxx - mnx, 1e-6f * (fabsf(mxx) + fabsf(mnx)) + 1e-30f);
        const float h = fmaxf(mxy - mny, 1e-6f * (fabsf(mxy) + fabsf(mny)) + 1e-30f);
        int target = M / 2;
        target = target < 1 ? 1 : target > kMaxCells ? kMaxCells : target;
        const float side = sqrtf(w * h / (float)target);
        float fx = ceilf(w / side), fy = ceilf(h / side);
        fx = fminf(fmaxf(fx, 1.f), (float)kMaxCells), fy = fminf(fmaxf(fy, 1.f), (float)kMaxCells);
        int gx = (int)fx, gy = (int)fy;
        if ((int64_t)gx * gy > kMaxCells) {  // (a very elongated domain)
            const float f = sqrtf((float)kMaxCells / ((float)gx * (float)gy));
            gx = (int)((float)gx * f), gy = (int)((float)gy * f);
            gx = gx < 1 ? 1 : gx, gy = gy < 1 ? 1 : gy;
            while ((int64_t)gx * gy > kMaxCells) gx > gy ? --gx : --gy;
        }
        hdr.x0 = mnx, hdr.y0 = mny, hdr.inv_cw = (float)gx / w, hdr.inv_ch = (float)gy / h, hdr.gx = gx, hdr.gy = gy;
    }
    __syncthreads();
    const float x0 = hdr.x0, y0 = hdr.y0, icw = hdr.inv_cw, ich = hdr.inv_ch;
    const int gx = hdr.gx, gy = hdr.gy, ncells = gx * gy;

    // 2. histogram
#pragma unroll
    for (int j = 0; j < kHold; ++j)
        if (finite2(hx[j], hy[j])) atomicAdd(&cnt[cell_coord(hy[j], y0, ich, gy) * gx + cell_coord(hx[j], x0, icw, gx)], 1);
    for (int i = tid + kHold * kBuildThreads; i < M; i += kBuildThreads) {
        const float x = inp[(int64_t)i * in_sn], y = inp[(int64_t)i * in_sn + in_sd];
        if (finite2(x, y)) atomicAdd(&cnt[cell_coord(y, y0, ich, gy) * gx + cell_coord(x, x0, icw, gx)], 1);
    }
    __syncthreads();

    // 3. exclusive scan of cnt[0 .. ncells): every thread owns kPer consecutive counters
    constexpr int kPer = (kMaxCells + kBuildThreads - 1) / kBuildThreads;
    int local[kPer], sum = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int c = tid * kPer + j;
        local[j] = c < ncells ? cnt[c] : 0;
        sum += local[j];
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < RPE_WAVE; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == RPE_WAVE - 1) wsum[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    int run = before + incl - sum;
    __syncthreads();  // (every thread has read its counters)
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int c = tid * kPer + j;
        if (c < ncells) cnt[c] = run, starts[c] = run;
        run += local[j];
    }
    int n_finite = 0;
    for (int w = 0; w < kBuildThreads / RPE_WAVE; ++w) n_finite += wsum[w];
    if (tid == 0) {
        starts[ncells] = n_finite;
        hdr.n_finite = n_finite, hdr.pad = 0;
        *(BinHeader *)base = hdr;
    }
    for (int i = n_finite + tid; i < M + kPadRecords; i += kBuildThreads)  // never-taken records behind the last one
        recs[i] = make_float4(0.f, 0.f, INFINITY, __int_as_float(0x7fffffff));
    __syncthreads();

    // 4. scatter (cnt[c] is now the fill pointer of cell c); the order inside a cell does not matter: the search compares indices
#pragma unroll
    for (int j = 0; j < kHold; ++j) {
        if (finite2(hx[j], hy[j])) {
            const float p[3] = {hx[j], hy[j], 0.f};
            const int pos = atomicAdd(&cnt[cell_coord(p[1], y0, ich, gy) * gx + cell_coord(p[0], x0, icw, gx)], 1);
            recs[pos] = make_float4(p[0], p[1], rpe_sqnorm<2>(p), __int_as_float(tid + j * kBuildThreads));
        }
    }
    for (int i = tid + kHold * kBuildThreads; i < M; i += kBuildThreads) {
        float p[3];
        p[0] = inp[(int64_t)i * in_sn], p[1] = inp[(int64_t)i * in_sn + in_sd], p[2] = 0.f;
        if (finite2(p[0], p[1])) {
            const int pos = atomicAdd(&cnt[cell_coord(p[1], y0, ich, gy) * gx + cell_coord(p[0], x0, icw, gx)], 1);
            recs[pos] = make_float4(p[0], p[1], rpe_sqnorm<2>(p), __int_as_float(i));
        }
    }
}

struct Best {
    float d;
    int i;
};

// minimum / maximum over each 16-lane row, left in every lane of the row: four rotate-and-combine steps on the DPP path
// (__shfl_xor is a ds_bpermute round trip per step, and fminf / fmaxf add a canonicalising v_max per operand).  Two wait
// states between a VALU write and a DPP read of the same register.
__device__ __forceinline__ float row_min16(float v) {
    asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}
__device__ __forceinline__ float row_max16(float v) {
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}
// ... and over the wave's four rows (values differ between the rows): wave-uniform result
__device__ __forceinline__ float wave_max_rows(float v) {
    v = row_max16(v);
    const float a = rpe_readlane(v, 0), b = rpe_readlane(v, 16), c = rpe_readlane(v, 32), d = rpe_readlane(v, 48);
    return rpe_uniform(fmaxf(fmaxf(a, b), fmaxf(c, d)));
}

__device__ __forceinline__ void take_point(const float4 &r, const float (&qm2)[3], float qq, Best &best) {
    const float p[3] = {r.x, r.y, 0.f};
    const float d = rpe_pair_dist<2>(qm2, qq, p, r.z);
    const int pi = __float_as_int(r.w);
    const bool take = (d < best.d) | ((d == best.d) & (pi < best.i));  // (bitwise: no short-circuit branches)
    best.d = take ? d : best.d;
    best.i = take ? pi : best.i;
}

// A wave owns QPW queries; lane (g, c) = (lane / QPW, lane % QPW) evaluates query c against every SL-th record of a range,
// SL = 64 / QPW slices: fewer queries a wave mean a smaller query box (fewer candidates per query) and SL-fold fewer
// dependent iterations; the slices of a query are merged at the end by (distance, index).
//
// The cells [ax, bx] x [ay, by]: the records of a cell row's run are one contiguous range; the ranges of up to 64 rows are
// fetched with one vector load each side.  Rows are then taken kRows at a time: the first kUnroll * SL records of each of them
// are requested TOGETHER (a wave's time is its chain of dependent memory round trips, and the usual rectangle is three rows of
// ~14 records: one round trip instead of three), the rest of a longer row follows in a loop.  No tail test: what lies behind
// a range are other cells' points or the +inf padding -- more candidates never hurt -- and the read-ahead stays inside
// kPadRecords.
constexpr int kUnroll = 4;
constexpr int kRows = 3;

template <int QPW>
__device__ __forceinline__ void visit_rect(const float4 *__restrict__ recs, const int *__restrict__ starts, int gx, int ax, int bx, int ay,
                                           int by, int lane, const float (&qm2)[3], float qq, Best &best) {
    constexpr int SL = RPE_WAVE / QPW;
    static_assert(kUnroll * SL <= kPadRecords, "read-ahead beyond the padding");
    if (ax > bx || ay > by) return;
    const unsigned goff = (unsigned)(lane / QPW) * 16u;  // this lane's slice, in bytes: loads are (uniform base) + goff + constant
    for (int r0 = ay; r0 <= by; r0 += RPE_WAVE) {
        const int nr = min(RPE_WAVE, by - r0 + 1);
        int rs = 0, re = 0;
        if (lane < nr) {
            const int row = (r0 + lane) * gx;
            rs = starts[row + ax], re = starts[row + bx + 1];
        }
        for (int r = 0; r < nr; r += kRows) {
            int s[kRows], e[kRows];
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                const int rr = min(r + j, nr - 1);  // (past the last row: that row again)
                s[j] = rpe_readlane(rs, rr), e[j] = rpe_readlane(re, rr);
            }
            float4 rec[kRows][kUnroll];
#pragma unroll
            for (int j = 0; j < kRows; ++j)
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) rec[j][u] = *(const float4 *)((const char *)(recs + s[j]) + goff + u * SL * 16);
#pragma unroll
            for (int j = 0; j < kRows; ++j)
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) take_point(rec[j][u], qm2, qq, best);
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                if (r + j >= nr) break;
                for (int i = s[j] + kUnroll * SL; i < e[j]; i += kUnroll * SL) {
                    float4 more[kUnroll];
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) more[u] = *(const float4 *)((const char *)(recs + i) + goff + u * SL * 16);
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) take_point(more[u], qm2, qq, best);
                }
            }
        }
    }
}

struct Rect {
    int ax, bx, ay, by;  // inclusive cell ranges
};

// the cells of `now` that are not in `old` (old inside now, or old.ax > old.bx: nothing visited yet): up to four rectangles
template <int QPW>
__device__ __forceinline__ void visit_new(const float4 *__restrict__ recs, const int *__restrict__ starts, int gx, Rect now, Rect old, int lane,
                                          const float (&qm2)[3], float qq, Best &best) {
    const bool none = old.ax > old.bx;
    const int n = none ? 1 : 4;
    for (int t = 0; t < n; ++t) {  // (not unrolled: one copy of the visiting code)
        Rect r = now;                                                 // t = 0, nothing visited yet: everything; else the rows above
        if (!none) {
            if (t == 0) r.by = old.ay - 1;
            else if (t == 1) r.ay = old.by + 1;                       // the rows below
            else if (t == 2) r.ay = old.ay, r.by = old.by, r.bx = old.ax - 1;  // left of the old rows
            else r.ay = old.ay, r.by = old.by, r.ax = old.bx + 1;     // right of them
        }
        visit_rect<QPW>(recs, starts, gx, r.ax, r.bx, r.ay, r.by, lane, qm2, qq, best);
    }
}

template <int QPW>
__global__ __launch_bounds__(256) void nearest2d_search_kernel(const float *__restrict__ qry, int64_t q_sb, int64_t q_sn, int64_t q_sd, int Q,
                                                               const char *__restrict__ ws, int64_t ws_stride, int64_t *__restrict__ idx,
                                                               float *__restrict__ dist) {
    const int lane = rpe_lane(), c = lane % QPW;
    const int wave = rpe_uniform((int)(threadIdx.x >> 6));
    const int b = blockIdx.y;
    const int qbase = (blockIdx.x * 4 + wave) * QPW;
    if (qbase >= Q) return;
    qry += (int64_t)b * q_sb;
    const char *base = ws + (int64_t)b * ws_stride;
    const BinHeader *hdr = (const BinHeader *)base;
    const int *starts = (const int *)(base + kHeaderBytes);
    const float4 *recs = (const float4 *)(base + kHeaderBytes + kStartsBytes);

    const int qi = min(qbase + c, Q - 1);
    float q[3];
    q[0] = qry[(int64_t)qi * q_sn], q[1] = qry[(int64_t)qi * q_sn + q_sd], q[2] = 0.f;
    const bool fin = finite2(q[0], q[1]);
    const float qq = rpe_sqnorm<2>(q);
    const float qm2[3] = {-2.0f * q[0], -2.0f * q[1], 0.f};
    static_assert(QPW == 16, "the query box is reduced over 16-lane DPP rows");
    // (every row of 16 lanes holds the same 16 queries)
    const float qminx = rpe_uniform(row_min16(fin ? q[0] : INFINITY)), qmaxx = rpe_uniform(row_max16(fin ? q[0] : -INFINITY));
    const float qminy = rpe_uniform(row_min16(fin ? q[1] : INFINITY)), qmaxy = rpe_uniform(row_max16(fin ? q[1] : -INFINITY));
    const float qqmax = rpe_uniform(row_max16(fin ? qq : 0.f));

    // (index -1: a +inf distance can then never be taken through the equality branch -- the sweeping kernels' strict `<` never
    // takes one either -- and a query that took nothing returns index 0 as theirs does)
    Best best{INFINITY, -1};
    if (qminx <= qmaxx) {  // (a wave of non-finite queries: every distance is NaN or +inf, nothing is ever taken)
        const float x0 = hdr->x0, y0 = hdr->y0, icw = hdr->inv_cw, ich = hdr->inv_ch;
        const int gx = hdr->gx, gy = hdr->gy;
        Rect old{1, 0, 1, 0};
        Rect now{max(cell_coord(qminx, x0, icw, gx) - 1, 0), min(cell_coord(qmaxx, x0, icw, gx) + 1, gx - 1),
                 max(cell_coord(qminy, y0, ich, gy) - 1, 0), min(cell_coord(qmaxy, y0, ich, gy) + 1, gy - 1)};
        bool last = false;
        for (;;) {
            visit_new<QPW>(recs, starts, gx, now, old, lane, qm2, qq, best);
            if (last) break;  // the bests only fell: nothing outside `now` can win any more
            old = now;
            const float worst = wave_max_rows(fin ? best.d : -INFINITY);  // (over every slice of every query)
            if (worst == INFINITY) {  // some lane has seen no point yet: widen geometrically
                if (now.ax == 0 && now.ay == 0 && now.bx == gx - 1 && now.by == gy - 1) break;
                const int wx = now.bx - now.ax + 1, wy = now.by - now.ay + 1;
                now = Rect{max(now.ax - wx, 0), min(now.bx + wx, gx - 1), max(now.ay - wy, 0), min(now.by + wy, gy - 1)};
                continue;
            }
            // every point that can still win or tie lies within R of its query (header comment)
            const float r2 = fmaxf(worst + 7.6293945e-6f * qqmax, 0.f) * 1.0000153f;
            const float r = sqrtf(r2) * 1.000001f + 1e-18f;  // (the floor covers products that underflow)
            const float lox = (qminx - r) - 2.4e-7f * (fabsf(qminx) + r), hix = (qmaxx + r) + 2.4e-7f * (fabsf(qmaxx) + r);
            const float loy = (qminy - r) - 2.4e-7f * (fabsf(qminy) + r), hiy = (qmaxy + r) + 2.4e-7f * (fabsf(qmaxy) + r);
            now = Rect{min(old.ax, cell_coord(lox, x0, icw, gx)), max(old.bx, cell_coord(hix, x0, icw, gx)),
                       min(old.ay, cell_coord(loy, y0, ich, gy)), max(old.by, cell_coord(hiy, y0, ich, gy))};
            if (now.ax == old.ax && now.bx == old.bx && now.ay == old.ay && now.by == old.by) break;
            last = true;
        }
    }
#pragma unroll
    for (int off = QPW; off < RPE_WAVE; off <<= 1) {  // the slices of a query
        const float od = __shfl_xor(best.d, off);
        const int oi = __shfl_xor(best.i, off);
        const bool take = (od < best.d) | ((od == best.d) & (oi < best.i));
        best.d = take ? od : best.d;
        best.i = take ? oi : best.i;
    }
    if (lane < QPW && qbase + lane < Q) {
        const int64_t o = (int64_t)b * Q + qbase + lane;
        idx[o] = best.i < 0 ? 0 : (int64_t)best.i;
        if (dist) dist[o] = best.d;
    }
}

}  // namespace

int64_t rpe_nearest2d_workspace_bytes(int B, int M) {
    if (B < 0 || M < 1) return 0;
    return ((int64_t)B * ws_stride_bytes(M) + 15) & ~15ll;
}

// (arguments checked by rpe_knn_multi; the workspace holds rpe_nearest2d_workspace_bytes(B, M) bytes, 16-byte aligned)
int rpe_nearest2d(const float *input, int64_t in_sb, int64_t in_sn, int64_t in_sd, const float *query, int64_t q_sb, int64_t q_sn,
                  int64_t q_sd, int B, int M, int Q, int64_t *idx, float *dist, void *workspace, hipStream_t st) {
    if (B == 0 || Q == 0) return 0;
    const int64_t stride = ws_stride_bytes(M);
    hipLaunchKernelGGL(nearest2d_build_kernel, dim3(B), dim3(kBuildThreads), 0, st, input, in_sb, in_sn, in_sd, query, q_sb, q_sn, q_sd, M, Q,
                       (char *)workspace, stride);
    int rc = rpe_launch_status();
    if (rc) return rc;
    // 16 queries a wave, each on four lanes: a wave of 16 meets ~3x fewer candidates per query than a wave of 64 would, and
    // there are 4x more waves to hide the memory round trips behind (measured: 22 / 40 / 46 us for 16 / 32 / 64 at level 1)
    constexpr int qpw = 16;
    hipLaunchKernelGGL(nearest2d_search_kernel<qpw>, dim3((Q + 4 * qpw - 1) / (4 * qpw), B), dim3(256), 0, st, query, q_sb, q_sn, q_sd, Q,
                       (const char *)workspace, stride, idx, dist);
    return rpe_launch_status();
}
