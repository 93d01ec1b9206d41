// k_nearest_neighbor on spatially ordered clouds (included by knn.hip, inside its anonymous namespace, after the matrix
// kernel: it shares that kernel's distance tiles, candidate lists, ranking and tie machinery).
//
// knn_mfma_kernel computes ALL Q x M distances twice (bound pass, collect pass): 214 us for 8 x (8192 -> 4096), k = 16, of
// which the matrix pipe needs 39 -- the rest is the per-step selection work of 256 steps per wave.  Here the cloud and the
// queries are first put into Morton-cell order (knn_grid_build_kernel: counting sort by cell of a 4096-cell grid over the
// set's bounding box; original indices carried as payload; a bounding box per 64-point step).  A wave's 16 queries are
// then neighbours in space, a step's 64 points too, and a step whose box is farther from the wave's query box than the
// current bound cannot hold a candidate for any of the 16 queries: it is skipped without computing a single distance.
// A wave touches ~10-25 of the 128 steps of an 8192-point cloud.
//
// Exactness.  The result is the k smallest of the SAME fp32 distances fl(fl(-2 q.p + |q|^2) + |p|^2) ranked by (distance,
// original index) -- sweep order does not enter -- followed by the same tie check / libstdc++ restatement (on the
// ORIGINAL cloud, in index order) as the other kernels, so indices equal torch.topk's position for position.  What must be
// shown is that no skipped point could have been collected, i.e. computed distance >= bound for all of a skipped step's
// points.  The box distance lb bounds the TRUE squared distance from below; the computed distance differs from the true
// one by at most 15 * 2^-24 * (|q|^2 + |p|^2) (three fmas, two adds, the rounded norms; 2 |q||p| <= |q|^2 + |p|^2), and
// the fp32 evaluation of lb itself by a relative 5 * 2^-24.  A step is skipped only if
//     lb * (1 - 2^-20) - 2^-19 * (max |q|^2 of the wave + max |p|^2 of the step)  >=  bound,
// 4x the worst-case error.  tests/test_gpu_ops.py runs the clouds of every KNN golden through this path.
//
// Per wave (16 queries, distances on the matrix pipe exactly as MfmaSweep::step):
//   lower bounds   lane l <- steps l, l + 64, ...: box-to-box distance with the margin above ("safe lower bound")
//   A0             the n0 nearest steps (2 for kk <= 4, else 4): running minima of 32 disjoint subsets per query (8
//                  registers x 4 lanes), bitonic sort of the 32 -> tau0 = their kk-th smallest: an upper bound of the
//                  query's kk-th distance from 128 / 256 nearby points
//   A'             every other step whose safe lower bound is below max_c tau0: the running minima continue -> tau, as tight
//                  as a sweep of the whole cloud gives it (all true neighbours lie in those steps)
//   B              every step below max_c tau: collect the points below the query's tau into the lane lists (positions in
//                  the sorted cloud), translate positions to original indices, rank by (distance, index), finish_query.
// Steps travel global -> LDS by LDS-DMA into a wave-private 4-slot ring, two steps ahead, counted vmcnt; no block barrier.

#ifndef RPE_KNN_GRID_PROBE
#define RPE_KNN_GRID_PROBE 0
#endif
constexpr int kGridCells = 4096;          // 12 Morton bits: 4 per dimension (D = 3), 6 (D = 2), 12 (D = 1)
constexpr int kGridBuildThreads = 1024;
constexpr int kGridMaxSS = 4;             // lower bounds live in registers, one step per lane and register: M <= 64 * 64 * 4
constexpr int kGridMaxM = 64 * 64 * kGridMaxSS;
constexpr int kGridBoxFloats = 8;         // min xyz, max xyz, max |p|^2, pad
constexpr int kGridSlots = 8;             // ring depth in steps
constexpr int kGridLead = 6;              // a step is requested this many steps before it is used (~2.5 k cycles of L2 latency against ~0.5 k of work per step)
constexpr int kGridSlotFloats = 256;      // a step in memory and in LDS alike: three coordinate rows of 64 + the |p|^2 row (layout below)

// A step (64 consecutive points of the sorted set) is ONE 1-KiB record [4][64]: rows 0-2 the coordinates, row 3 |p|^2 (+inf in
// the padding).  Coordinate row g is rotated by 16 g floats -- point i sits at ((i + 16 g) & 63) -- so that the A-fragment
// reads of the distance tiles (lane (g, c) reads point 16 j + c of row g) fall into different LDS banks for g = 0, 1, 2
// without padding, and the record is the unit of the LDS-DMA: one global_load_lds_dwordx4 per step and wave.  (Four dword
// DMAs per step, as the first version issued them, ran at one instruction per ~90 cycles and CU: 2900 cycles a step.)
__device__ __forceinline__ int grid_at(int row, int i) { return row * 64 + ((i + 16 * row) & 63); }

struct rpe_grid_set {  // one point set in grid order (rpe_knn_grid_build)
    const float *sorted;   // [B][Npad / 64][4][64]: the step records
    const int *perm;       // [B][Npad]: original index of sorted position
    const float *boxes;    // [B][Npad / 64 + 1][8]: per step; the last entry is the whole set
};

__device__ __forceinline__ unsigned grid_spread3(unsigned v) {  // 4 bits -> every third bit
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}
__device__ __forceinline__ unsigned grid_spread2(unsigned v) {  // 6 bits -> every second bit
    v = (v | (v << 4)) & 0x0303u;   // ..54 ..... ..3210 -> split
    v = (v | (v << 2)) & 0x0333u;
    v = (v | (v << 1)) & 0x0555u;
    return v;
}
template <int D>
__device__ __forceinline__ int grid_cell(const float (&p)[3], const float (&lo)[3], const float (&scale)[3]) {
    constexpr int bits = D == 3 ? 4 : D == 2 ? 6 : 12;
    constexpr float top = (float)((1 << bits) - 1);
    unsigned c[3];
#pragma unroll
    for (int d = 0; d < D; ++d) c[d] = (unsigned)fminf(fmaxf((p[d] - lo[d]) * scale[d], 0.f), top);  // (NaN -> 0)
    if (D == 3) return (int)(grid_spread3(c[0]) | (grid_spread3(c[1]) << 1) | (grid_spread3(c[2]) << 2));
    if (D == 2) return (int)(grid_spread2(c[0]) | (grid_spread2(c[1]) << 1));
    return (int)c[0];
}

struct GridBuildJob {
    const float *pts;
    int64_t sb, sn, sd;
    int N;
    float *sorted;
    int *perm;
    float *boxes;
};
struct GridBuildJobs {
    GridBuildJob job[RPE_KNN_MAX_JOBS];
};

// One workgroup per (set, batch element).
template <int D>
__global__ __launch_bounds__(kGridBuildThreads) void knn_grid_build_kernel(GridBuildJobs jobs) {
    const GridBuildJob &J = jobs.job[blockIdx.y];
    const int b = blockIdx.x, N = J.N, Npad = (N + 63) & ~63, steps = Npad >> 6;
    const float *pts = J.pts + (int64_t)b * J.sb;
    float *sorted = J.sorted + (int64_t)b * 4 * Npad;
    int *perm = J.perm + (int64_t)b * Npad;
    float *boxes = J.boxes + (int64_t)b * (steps + 1) * kGridBoxFloats;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kGridBuildThreads / RPE_WAVE;
    __shared__ int hist[kGridCells];
    __shared__ float red[kWaves][8];
    __shared__ int wsum[kWaves];

    // ---- bounding box of the set (and its largest |p|^2)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, ppmax = 0.f;
    for (int i = tid; i < N; i += kGridBuildThreads) {
        float p[3];
        load_point<D>(pts, J.sn, J.sd, i, p);
#pragma unroll
        for (int d = 0; d < D; ++d) lo[d] = fminf(lo[d], p[d]), hi[d] = fmaxf(hi[d], p[d]);
        ppmax = fmaxf(ppmax, rpe_sqnorm<D>(p));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int d = 0; d < D; ++d) lo[d] = fminf(lo[d], __shfl_xor(lo[d], off)), hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off));
        ppmax = fmaxf(ppmax, __shfl_xor(ppmax, off));
    }
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) red[wave][d] = lo[d], red[wave][3 + d] = hi[d];
        red[wave][6] = ppmax;
    }
    for (int c = tid; c < kGridCells; c += kGridBuildThreads) hist[c] = 0;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; ++d) lo[d] = red[0][d], hi[d] = red[0][3 + d];
    ppmax = red[0][6];
    for (int w = 1; w < kWaves; ++w) {
#pragma unroll
        for (int d = 0; d < 3; ++d) lo[d] = fminf(lo[d], red[w][d]), hi[d] = fmaxf(hi[d], red[w][3 + d]);
        ppmax = fmaxf(ppmax, red[w][6]);
    }
    constexpr int bits = D == 3 ? 4 : D == 2 ? 6 : 12;
    float scale[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < D; ++d) scale[d] = hi[d] > lo[d] ? (float)(1 << bits) / (hi[d] - lo[d]) : 0.f;

    // ---- counting sort by cell: histogram, exclusive scan (4 cells a thread), scatter
    for (int i = tid; i < N; i += kGridBuildThreads) {
        float p[3];
        load_point<D>(pts, J.sn, J.sd, i, p);
        atomicAdd(&hist[grid_cell<D>(p, lo, scale)], 1);
    }
    __syncthreads();
    constexpr int per = kGridCells / kGridBuildThreads;  // 4
    int cnt[per], mine = 0;
#pragma unroll
    for (int j = 0; j < per; ++j) cnt[j] = hist[tid * per + j], mine += cnt[j];
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off);
        incl += lane >= off ? up : 0;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = incl - mine;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    int start[per];
#pragma unroll
    for (int j = 0; j < per; ++j) start[j] = before, hist[tid * per + j] = before, before += cnt[j];
    __syncthreads();
    for (int i = tid; i < N; i += kGridBuildThreads) {
        float p[3];
        load_point<D>(pts, J.sn, J.sd, i, p);
        perm[atomicAdd(&hist[grid_cell<D>(p, lo, scale)], 1)] = i;
    }
    __syncthreads();
    // the atomics hand out a cell's slots in arrival order: put every (small) cell back into index order, so that the
    // layout -- and with it the step boxes and the kernel's timing -- does not depend on the scheduling of this launch
#pragma unroll
    for (int j = 0; j < per; ++j) {
        const int s = start[j], n = cnt[j];
        if (n < 2 || n > 48) continue;
        for (int a = 1; a < n; ++a) {
            const int v = perm[s + a];
            int t = a - 1;
            while (t >= 0 && perm[s + t] > v) perm[s + t + 1] = perm[s + t], --t;
            perm[s + t + 1] = v;
        }
    }
    __syncthreads();

    // ---- the set in grid order (a wave owns one 64-point step per round), the step boxes
    for (int pos = tid; pos < Npad; pos += kGridBuildThreads) {
        const bool valid = pos < N;
        const int i = valid ? perm[pos] : 0;
        float p[3] = {0.f, 0.f, 0.f};
        if (valid) load_point<D>(pts, J.sn, J.sd, i, p);
        const float pp = valid ? rpe_sqnorm<D>(p) : INFINITY;
        float *rec = sorted + (int64_t)(pos >> 6) * kGridSlotFloats;
#pragma unroll
        for (int d = 0; d < 3; ++d) rec[grid_at(d, pos & 63)] = p[d];  // (missing dimensions: zero)
        rec[3 * 64 + (pos & 63)] = pp;
        if (!valid) perm[pos] = 0;
        float bl[3], bh[3], bp = valid ? pp : 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) bl[d] = valid ? p[d] : INFINITY, bh[d] = valid ? p[d] : -INFINITY;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int d = 0; d < 3; ++d) bl[d] = fminf(bl[d], __shfl_xor(bl[d], off)), bh[d] = fmaxf(bh[d], __shfl_xor(bh[d], off));
            bp = fmaxf(bp, __shfl_xor(bp, off));
        }
        if (lane == 0) {
            float *box = boxes + (int64_t)(pos >> 6) * kGridBoxFloats;
#pragma unroll
            for (int d = 0; d < 3; ++d) box[d] = bl[d], box[3 + d] = bh[d];
            box[6] = bp, box[7] = 0.f;
        }
    }
    if (tid == 0) {
        float *box = boxes + (int64_t)steps * kGridBoxFloats;
#pragma unroll
        for (int d = 0; d < 3; ++d) box[d] = lo[d], box[3 + d] = hi[d];
        box[6] = ppmax, box[7] = 0.f;
    }
}

// ---- the search ------------------------------------------------------------------------------------------------------------
// Diagnostics (rpe_knn_grid_set_stats; null in normal use): per launch group the kernel adds, per wave, [0] 1, [1-3] steps
// evaluated in A0 / A' / B, [4] candidates collected, [5] queries sent to the serial sweep, [6] queries with equal distances,
// [7-12] clock cycles (s_memtime) spent in: lower bounds, A0 + its bound, A' + its bound, B, translate + rank, finish.
__device__ unsigned long long *g_grid_stats = nullptr;

struct GridJob {
    rpe_knn_job j;          // the ORIGINAL arrays (tie restatement, serial fallback) and the outputs
    rpe_grid_set in, q;     // the same sets in grid order
};
struct GridJobs {
    GridJob job[RPE_KNN_MAX_JOBS];
};

struct GridWaveLds {
    float ring[kGridSlots][kGridSlotFloats];
    float ones[16];              // the constant k-slot of the distance product (directly behind the ring: GridSweep::init)
    int steps[kGridMaxSS * 64];  // the steps of the current pass, in order
    int count[RPE_WAVE];
    unsigned long long list[kLaneList][RPE_WAVE];  // as MfmaBlockLds::list
};
struct GridBlockLds {
    GridWaveLds w[kWavesPerBlock];
};

// bitonic sort, ascending, of the 32 values {v[i] of lanes c, 16 + c, 32 + c, 48 + c}: element e = 8 g + i, all 16 c at once
__device__ __forceinline__ void bitonic32_cols(float (&v)[8], int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int s = 2; s <= 32; s <<= 1) {
#pragma unroll
        for (int j = s >> 1; j > 0; j >>= 1) {
            if (j < 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i & j) continue;
                    const bool asc = s < 8 ? (i & s) == 0 : s == 8 ? (g & 1) == 0 : s == 16 ? (g & 2) == 0 : true;
                    float x = v[i], y = v[i | j];
                    knn_cex(x, y, asc);
                    v[i] = x;
                    v[i | j] = y;
                }
            } else {  // partner lane l ^ 2 j (j = 8: g ^ 1, j = 16: g ^ 2)
                const bool lower = (lane & (2 * j)) == 0;
                const bool asc = s == 16 ? (g & 2) == 0 : true;
                const bool keep_min = lower == asc;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float x = v[i];
                    const float y = __shfl_xor(x, 2 * j);
                    v[i] = ((y < x) == keep_min) ? y : x;
                }
            }
        }
    }
}

// the kk-th smallest (kk <= 32) of the wave's 32 running minima per query, as a strict bound: the next float up
__device__ __forceinline__ float grid_bound(const float (&lm)[8], int kk, int lane) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lm[i];
    bitonic32_cols(v, lane);
    const int ei = (kk - 1) & 7;
    float sel = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) sel = ei == i ? v[i] : sel;
    const float bound = __shfl(sel, (((kk - 1) >> 3) << 4) | (lane & 15));
    const int bits = __float_as_int(bound);
    return !(bound < INFINITY) ? INFINITY : bound == 0.f ? __int_as_float(1) : bound > 0.f ? __int_as_float(bits + 1) : __int_as_float(bits - 1);
}

__device__ __forceinline__ float grid_wave_max16(float v) {  // max over the 16 queries (lanes c of every group hold the same value)
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return rpe_uniform(v);
}

template <int D>
struct GridSweep {
    const float *sorted;  // this cloud's step records
    int lane;
    float *ring;          // this wave's [kGridSlots][kGridSlotFloats]
    float qb;
    int a_off[4];         // floats: where lane (g, c) finds its A operand of tile j in a record (g == 3: the row of ones behind the ring)
    __device__ __forceinline__ void init(int lane_, float *ring_, int ones_off) {
        lane = lane_, ring = ring_;
        const int g = lane >> 4, c = lane & 15;
#pragma unroll
        for (int j = 0; j < 4; ++j) a_off[j] = g < D ? grid_at(g, 16 * j + c) : ones_off + c;
    }
    // ONE DMA per request: the counted waits below rely on it
    __device__ __forceinline__ void request(int step, int slot) const {
#if RPE_KNN_GRID_PROBE == 1  // timing only: no DMA at all (stale LDS contents)
        return;
#endif
        const float *src = sorted + (int64_t)step * kGridSlotFloats + 4 * lane;
        __builtin_amdgcn_global_load_lds((knn_glb_void_t *)src, (knn_lds_void_t *)(ring + slot * kGridSlotFloats), 16, 0, 0);
    }
    // distances of the 64 points in `slot` to the 16 queries: register 4 b + r of lane (g, c) = point 16 b + 4 g + r, query c
    __device__ __forceinline__ knn_f32x16 step(int slot) const {
        const float *s = ring + slot * kGridSlotFloats;
        const float pp = s[3 * 64 + lane];  // (+inf in the padding: those distances are +inf)
        const float *one_or = D > (lane >> 4) ? s : ring;  // lanes of the constant k-slot read the ones behind the ring, whatever the slot
        const knn_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const knn_f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(one_or[a_off[0]], qb, zero, 0, 0, 0);
        const knn_f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(one_or[a_off[1]], qb, zero, 0, 0, 0);
        const knn_f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(one_or[a_off[2]], qb, zero, 0, 0, 0);
        const knn_f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(one_or[a_off[3]], qb, zero, 0, 0, 0);
        const knn_f32x8 d01 = __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7);
        const knn_f32x8 d23 = __builtin_shufflevector(d2, d3, 0, 1, 2, 3, 4, 5, 6, 7);
        const knn_f32x16 acc = __builtin_shufflevector(d01, d23, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        return __builtin_amdgcn_mfma_f32_16x16x1f32(pp, 1.0f, acc, 0, 0, 0);
    }
    // f(step id, distances) for every step whose bit is set in mask[0 .. n_ss), in order.  The list goes to LDS first; step
    // n + kGridLead is requested before step n is used, and when the list runs out the requests repeat its last entry, so
    // every round waits for exactly vmcnt(kGridLead).
    template <class F>
    __device__ __forceinline__ void for_steps(const unsigned long long (&mask)[kGridMaxSS], int n_ss, int *list, F &&f) const {
        int total = 0;
#pragma unroll
        for (int ss = 0; ss < kGridMaxSS; ++ss) {
            const unsigned long long m = ss < n_ss ? mask[ss] : 0ull;
            if ((m >> lane) & 1ull) list[total + __popcll(m & ((1ull << lane) - 1ull))] = ss * 64 + lane;
            total += __popcll(m);
        }
        if (total == 0) return;
        const int last = total - 1;
#pragma unroll
        for (int i = 0; i < kGridLead; ++i) request(rpe_uniform(list[min(i, last)]), i);
        for (int n = 0; n < total; ++n) {
            request(rpe_uniform(list[min(n + kGridLead, last)]), (n + kGridLead) & (kGridSlots - 1));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kGridLead) : "memory");
#if RPE_KNN_GRID_PROBE == 2  // timing only: the DMAs and the waits without the distance tiles
            continue;
#endif
            f(rpe_uniform(list[n]), step(n & (kGridSlots - 1)));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing requests land before the ring is reused
    }
};

template <int D>
__global__ __launch_bounds__(kWavesPerBlock * RPE_WAVE) void knn_grid_kernel(GridJobs jobs, int k, int exact_ties) {
    extern __shared__ __align__(16) unsigned char grid_lds_raw[];
    GridBlockLds &lds = *reinterpret_cast<GridBlockLds *>(grid_lds_raw);
    const GridJob &G = jobs.job[blockIdx.z];
    const rpe_knn_job &J = G.j;
    const int M = J.M, Q = J.Q, Mpad = (M + 63) & ~63, Qpad = (Q + 63) & ~63, steps = Mpad >> 6;
    const int lane = rpe_lane(), g = lane >> 4, c = lane & 15;
    const int wave = rpe_uniform((int)(threadIdx.x >> 6));
    const int b = blockIdx.y;
    if ((int)blockIdx.x * kWavesPerBlock * kMq >= Q) return;  // (block-uniform)
    const int qbase = (blockIdx.x * kWavesPerBlock + wave) * kMq;  // position in the sorted query set
    const float *inp = J.input + (int64_t)b * J.in_sb;
    const float *in_sorted = G.in.sorted + (int64_t)b * 4 * Mpad;
    const int *in_perm = G.in.perm + (int64_t)b * Mpad;
    const float *in_boxes = G.in.boxes + (int64_t)b * (steps + 1) * kGridBoxFloats;
    const float *q_sorted = G.q.sorted + (int64_t)b * 4 * Qpad;
    const int *q_perm = G.q.perm + (int64_t)b * Qpad;
    int64_t *__restrict__ idx = J.idx;
    float *__restrict__ dist = J.dist;
    if ((threadIdx.x & 63) < 16) lds.w[threadIdx.x >> 6].ones[threadIdx.x & 63] = 1.0f;
    __shared__ TieQueue tq;
    if (threadIdx.x == 0) tq.n = 0;
    __syncthreads();
    GridWaveLds &L = lds.w[wave];
    const int kk = (exact_ties && k < M && k < RPE_WAVE) ? k + 1 : k;
    const int n_ss = (steps + 63) >> 6;

    unsigned long long *const stats = g_grid_stats;
#if RPE_KNN_GRID_PROBE == 3  // timing only: everything twice, the statistics of the SECOND pass (warm instruction cache, TLB, L2)
    for (int rep = 0; rep < 2; ++rep)
#endif
    if (qbase < Q) {  // (wave-uniform; a wave beyond Q only helps with the queued ties below)
#if RPE_KNN_GRID_PROBE == 3
        unsigned long long *const stats = rep == 1 ? g_grid_stats : nullptr;
#endif
        unsigned long long t_mark = stats ? __builtin_readcyclecounter() : 0ull;
        auto lap = [&](int slot) {
            if (stats) {
                const unsigned long long now = __builtin_readcyclecounter();
                if (lane == 0) atomicAdd(&stats[slot], now - t_mark);
                t_mark = now;
            }
        };
        auto count = [&](int slot, unsigned long long v) {
            if (stats && lane == 0) atomicAdd(&stats[slot], v);
        };
        count(0, 1);
        float qv[3] = {0.f, 0.f, 0.f};  // lane c of every group: the query at sorted position qbase + c
        {
            const int qp = min(qbase + c, Q - 1);
#pragma unroll
            for (int d = 0; d < D; ++d) qv[d] = q_sorted[(int64_t)(qp >> 6) * kGridSlotFloats + grid_at(d, qp & 63)];
        }
        const float qq_own = rpe_sqnorm<D>(qv);
        GridSweep<D> sw;
        sw.sorted = in_sorted;
        sw.init(lane, &L.ring[0][0], kGridSlots * kGridSlotFloats);
        sw.qb = g == 0 ? -2.0f * qv[0] : g == 1 ? -2.0f * qv[1] : g == 2 ? -2.0f * qv[2] : qq_own;

        // ---- box of the wave's queries; safe lower bound of every step (lane l: steps l, l + 64, ...)
        float qlo[3], qhi[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float a = qv[d], z = qv[d];
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) a = fminf(a, __shfl_xor(a, off)), z = fmaxf(z, __shfl_xor(z, off));
            qlo[d] = rpe_uniform(a), qhi[d] = rpe_uniform(z);
        }
        const float qq_max = grid_wave_max16(qq_own);
        float slb[kGridMaxSS];
#pragma unroll
        for (int ss = 0; ss < kGridMaxSS; ++ss) {
            slb[ss] = INFINITY;
            const int st = ss * 64 + lane;
            if (ss < n_ss && st < steps) {
                const float4 b0 = *reinterpret_cast<const float4 *>(in_boxes + (int64_t)st * kGridBoxFloats);
                const float4 b1 = *reinterpret_cast<const float4 *>(in_boxes + (int64_t)st * kGridBoxFloats + 4);
                const float blo[3] = {b0.x, b0.y, b0.z}, bhi[3] = {b0.w, b1.x, b1.y};
                float lb = 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float gap = fmaxf(fmaxf(blo[d] - qhi[d], qlo[d] - bhi[d]), 0.f);
                    lb = lb + gap * gap;
                }
                slb[ss] = lb * (1.0f - 0x1p-20f) - 0x1p-19f * (qq_max + b1.z);
            }
        }

        lap(7);
        // ---- A0: the nearest steps
        float lm[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) lm[i] = INFINITY;
        auto fold = [&](const knn_f32x16 &acc) {
#pragma unroll
            for (int i = 0; i < 8; ++i) lm[i] = knn_min3(lm[i], acc[i], acc[i + 8]);
        };
        unsigned long long seen[kGridMaxSS], act[kGridMaxSS];
#pragma unroll
        for (int ss = 0; ss < kGridMaxSS; ++ss) seen[ss] = 0ull;
        const int n0 = min(kk <= 4 ? 2 : 4, steps);
        for (int r = 0; r < n0; ++r) {
            float best = INFINITY;
            int best_ss = 0;
#pragma unroll
            for (int ss = 0; ss < kGridMaxSS; ++ss) {
                const bool taken = (seen[ss] >> lane) & 1ull;
                const float v = (ss < n_ss && ss * 64 + lane < steps && !taken) ? slb[ss] : INFINITY;
                if (v < best) best = v, best_ss = ss;
            }
            float wmin = best;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wmin = fminf(wmin, __shfl_xor(wmin, off));
            const unsigned long long who = __ballot(best == wmin && best < INFINITY);
            if (who == 0ull) break;
            const int l = __builtin_ctzll(who);
            const int ss = rpe_readlane(best_ss, l);
#pragma unroll
            for (int t = 0; t < kGridMaxSS; ++t) seen[t] |= t == ss ? 1ull << l : 0ull;
        }
        sw.for_steps(seen, n_ss, L.steps, [&](int, const knn_f32x16 &acc) { fold(acc); });
        const float tau0 = grid_wave_max16(grid_bound(lm, kk, lane));
        if (stats) {
            unsigned long long n = 0;
            for (int ss = 0; ss < kGridMaxSS; ++ss) n += __popcll(seen[ss]);
            count(1, n);
        }
        lap(8);

        // ---- A': the other steps that can hold one of the kk nearest of any of the 16 queries
#pragma unroll
        for (int ss = 0; ss < kGridMaxSS; ++ss) act[ss] = ss < n_ss ? (__ballot(slb[ss] < tau0) & ~seen[ss]) : 0ull;
        sw.for_steps(act, n_ss, L.steps, [&](int, const knn_f32x16 &acc) { fold(acc); });
        const float tau = grid_bound(lm, kk, lane);
        const float tau_max = grid_wave_max16(tau);
        if (stats) {
            unsigned long long n = 0;
            for (int ss = 0; ss < kGridMaxSS; ++ss) n += __popcll(act[ss]);
            count(2, n);
        }
        lap(9);

        // ---- B: collect (position in the sorted cloud, distance) below the query's bound, as knn_mfma_kernel does
        const unsigned list_base = (unsigned)(uintptr_t)&L.list[0][lane];
        int cnt = 0;
#pragma unroll
        for (int ss = 0; ss < kGridMaxSS; ++ss) act[ss] = ss < n_ss ? __ballot(slb[ss] < tau_max) : 0ull;
        sw.for_steps(act, n_ss, L.steps, [&](int st, const knn_f32x16 &acc) {
            const int base = st * RPE_WAVE + 4 * g;
            float m4[4];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) m4[bb] = knn_min(knn_min3(acc[4 * bb], acc[4 * bb + 1], acc[4 * bb + 2]), acc[4 * bb + 3]);
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                if (__ballot(m4[bb] < tau) == 0ull) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = acc[4 * bb + r];
                    const unsigned addr = list_base + (unsigned)min(cnt, kLaneList - 1) * (unsigned)(RPE_WAVE * 8);
                    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(addr), "v"(d), "v"(base + 16 * bb + r) : "memory");
                    cnt += d < tau ? 1 : 0;
                }
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        L.count[lane] = cnt;
        if (stats) {
            unsigned long long n = 0;
            for (int ss = 0; ss < kGridMaxSS; ++ss) n += __popcll(act[ss]);
            count(3, n);
            int tot = cnt;
            for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
            count(4, (unsigned long long)tot);
        }
        lap(10);

        // ---- positions -> original indices, then rank by (distance, index) and finish: as knn_mfma_kernel
        const int have = min(cnt, kLaneList - 1);
        int have_g[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) have_g[gg] = min(L.count[16 * gg + c], kLaneList - 1);
        const unsigned long long full_lanes = __ballot(cnt >= kLaneList);
        const int n_query = have_g[0] + have_g[1] + have_g[2] + have_g[3];
        int dense_at = 0;
#pragma unroll
        for (int gg = 0; gg < 3; ++gg) dense_at += gg < g ? have_g[gg] : 0;
        int max_n = n_query;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) max_n = max(max_n, __shfl_xor(max_n, off));
        max_n = rpe_uniform(max_n);
        unsigned long long *const dense = &L.list[0][0] + c * RPE_WAVE;
        {
            unsigned long long own[kLaneList - 1];
#pragma unroll
            for (int sl = 0; sl < kLaneList - 1; ++sl) own[sl] = L.list[sl][lane];
#pragma unroll
            for (int sl = 0; sl < kLaneList - 1; ++sl)
                if (sl < have) own[sl] = ((unsigned long long)(unsigned)in_perm[(int)(own[sl] >> 32)] << 32) | (own[sl] & 0xffffffffull);
#pragma unroll
            for (int sl = 0; sl < kLaneList - 1; ++sl)
                if (sl < have) dense[dense_at + sl] = own[sl];
        }
        auto key_of = [](unsigned long long e) {
            const unsigned db = (unsigned)e, ix = (unsigned)(e >> 32);
            const unsigned ord = db ^ (((int)db >> 31) | 0x80000000u);
            return ((unsigned long long)ord << 32) | ix;
        };
        auto rank_all = [&](auto ns_tag) {
            constexpr int NS = decltype(ns_tag)::value;
            unsigned long long mine_e[NS], mine_k[NS];
            int rank[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int at = g + 4 * j;
                mine_e[j] = dense[at & (RPE_WAVE - 1)];
                mine_k[j] = at < n_query ? key_of(mine_e[j]) : ~0ull;
                rank[j] = 0;
            }
            for (int o = 0; o < max_n; ++o) {
                const unsigned long long ok = o < n_query ? key_of(dense[o]) : ~0ull;
#pragma unroll
                for (int j = 0; j < NS; ++j) rank[j] += ok < mine_k[j] ? 1 : 0;
            }
#pragma unroll
            for (int j = 0; j < NS; ++j)
                if (g + 4 * j < n_query) dense[rank[j]] = mine_e[j];
        };
        if (max_n <= 16) rank_all(std::integral_constant<int, 4>{});
        else if (max_n <= 32) rank_all(std::integral_constant<int, 8>{});
        else rank_all(std::integral_constant<int, 16>{});

        lap(11);
        const int qo_all = q_perm[min(qbase + c, Q - 1)];  // the queries' ORIGINAL indices: one load, not one per query
        for (int q = 0; q < kMq; ++q) {
            if (qbase + q >= Q) break;
            const int qo = rpe_readlane(qo_all, q);
            float qc[3], qm2[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) qc[d] = rpe_readlane(qv[d], q);
            const float qq = rpe_sqnorm<D>(qc);
#pragma unroll
            for (int d = 0; d < 3; ++d) qm2[d] = -2.0f * qc[d];
            float Ld;
            int Li;
            const int n = rpe_readlane(n_query, q);
            if (((full_lanes >> q) & 0x0001000100010001ull) != 0ull || n > RPE_WAVE) {
                serial_select<D>(inp, J.in_sn, J.in_sd, M, qm2, qq, kk, lane, Ld, Li);
                count(5, 1);
            } else {
                const unsigned long long e = L.list[0][q * RPE_WAVE + lane];
                Ld = lane < n ? __int_as_float((int)(unsigned)e) : INFINITY;
                Li = lane < n ? (int)(e >> 32) : 0;
            }
            if (has_ties(Ld, k, kk, exact_ties, lane)) {
                if (lane == 0) tq.query[atomicAdd(&tq.n, 1)] = qo;
                count(6, 1);
            } else if (lane < k) {
                const int64_t o = ((int64_t)b * Q + qo) * k + lane;
                idx[o] = (int64_t)Li;
                if (dist) dist[o] = Ld;
            }
        }
        lap(12);
    }
    if (exact_ties) drain_ties<D>(&tq, J, b, k, lane, wave);
}
