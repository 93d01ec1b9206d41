// k_nearest_neighbor + squared_distance for gfx950.
//
// Replaces k_nearest_neighbor_{2d,3d}_kernel (k_nearest_neighbor_kernel.cu:8-112:
// one thread per query, 32-entry local arrays) with a wave-cooperative design:
//
//   * one wave owns QW queries; their coordinates are wave-uniform (SGPRs);
//   * the 64 lanes sweep the input cloud 64 points at a time (coalesced loads,
//     either point layout through strides);
//   * k == 1: every lane keeps its own running minimum, one butterfly at the end;
//   * k >= 2: the sorted top-k list of a query lives ACROSS the lanes (lane r holds
//     rank r, so k <= 64); a tile's candidates are found with one ballot against
//     the current k-th distance and inserted with a one-lane shift.
//
// Arithmetic and tie rule follow the CPU fallback (wrapper.py:40-52,115-117), see
// common.h; equal distances are ordered by input index.
#include <math.h>

#include "common.h"

namespace {

constexpr int kWavesPerBlock = 4;

template <int D>
__device__ __forceinline__ void load_point(const float *base, int64_t sn, int64_t sd, int i, float (&p)[3]) {
    const float *a = base + (int64_t)i * sn;
    p[0] = a[0];
    p[1] = D > 1 ? a[sd] : 0.f;
    p[2] = D > 2 ? a[2 * sd] : 0.f;
}

// The sweeps read the cloud 64 points at a time through a small wave-private LDS ring filled by LDS-DMA
// (global_load_lds_dword: no staging registers, so nothing of a tile in flight is loop-carried and hipcc has no reason to
// drain the memory queue at the loop head, which it does -- s_waitcnt vmcnt(0) -- for register prefetches carried
// around the back edge: that wait exposed one L2 round trip per tile, 46 % of the wave cycles of the k = 16 search).
// Tile t + 2 is requested before tile t is used; every request issues exactly D DMAs (lanes past the end re-read the
// last point), so the counted wait below is exact.  Nothing else in the sweeps touches vector memory.
typedef __attribute__((address_space(3))) void knn_lds_void_t;
typedef __attribute__((address_space(1))) const void knn_glb_void_t;
constexpr int kRingSlots = 4;

template <int D>
struct TileStream {
    const float *inp;
    int64_t sn, sd;
    int M, lane;
    float *ring;  // this wave's [kRingSlots][3][64] floats of LDS
    __device__ __forceinline__ void request(int base) const {
        const int pi = min(base + lane, M - 1);
        const float *src = inp + (int64_t)pi * sn;
        float *dst = ring + ((base >> 6) & (kRingSlots - 1)) * (3 * RPE_WAVE);
#pragma unroll
        for (int d = 0; d < D; ++d)
            __builtin_amdgcn_global_load_lds((knn_glb_void_t *)(src + d * sd), (knn_lds_void_t *)(dst + d * RPE_WAVE), 4, 0, 0);
    }
    __device__ __forceinline__ void start() const {
        request(0);
        request(RPE_WAVE);
    }
    // the points of tile `base` (lanes past the end hold the last point: callers mask them), and the request for tile base + 128
    __device__ __forceinline__ void next(int base, float (&p)[3]) const {
        request(base + 2 * RPE_WAVE);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * D) : "memory");  // tile `base` has landed (two younger requests may be pending)
        const float *src = ring + ((base >> 6) & (kRingSlots - 1)) * (3 * RPE_WAVE) + lane;
        p[0] = src[0];
        p[1] = D > 1 ? src[RPE_WAVE] : 0.f;
        p[2] = D > 2 ? src[2 * RPE_WAVE] : 0.f;
    }
    // after the last tile: the two trailing requests must land before the ring is reused or the kernel ends
    __device__ __forceinline__ void finish() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

template <int D, int QW>
struct Queries {
    float qm2[QW][3];
    float qq[QW];
    __device__ __forceinline__ void load(const float *qry, int64_t sn, int64_t sd, int qbase, int Q) {
#pragma unroll
        for (int j = 0; j < QW; ++j) {
            int qi = min(qbase + j, Q - 1);
            float q[3];
            load_point<D>(qry, sn, sd, qi, q);
#pragma unroll
            for (int d = 0; d < 3; ++d) q[d] = rpe_uniform(q[d]);
            qq[j] = rpe_sqnorm<D>(q);
#pragma unroll
            for (int d = 0; d < 3; ++d) qm2[j][d] = -2.0f * q[d];
        }
    }
};

// ---- equal distances exactly as the reference resolves them ---------------------------------------------------------
// The reference's k_nearest_neighbor is squared_distance + torch.topk (wrapper.py:115-117).  ATen's CPU topk runs
// std::partial_sort when k * 64 <= M, else std::nth_element + std::sort of the first k - 1; its comparator sees the
// value only, so which of several EQUAL distances is kept, and their order, is whatever libstdc++'s heap / introselect
// code does (aten/src/ATen/native/TopKImpl.h; bits/stl_heap.h, bits/stl_algo.h).  The sweep below keeps one entry more
// than k: any two equal neighbours among the k + 1 best mean the result could depend on that code, and only then the
// query is redone by a literal restatement of it -- heap_* on a lane-distributed heap for the partial_sort case,
// seq_* by one lane on an LDS copy of the row for the (small M) nth_element case.  k <= 63 (the reference's extension caps
// k at 32, k_nearest_neighbor_kernel.cu:24,68); NaN distances are not modelled (the sweep never selects them).
struct LaneHeap {  // element r of the heap lives in lane r
    float v;
    int i;
    __device__ __forceinline__ float val(int r) const { return rpe_readlane(v, r); }
    __device__ __forceinline__ void move(int dst, int src, int lane) {  // heap[dst] = heap[src]
        const float sv = rpe_readlane(v, src);
        const int si = rpe_readlane(i, src);
        v = lane == dst ? sv : v;
        i = lane == dst ? si : i;
    }
    __device__ __forceinline__ void set(int dst, float nv, int ni, int lane) {
        v = lane == dst ? nv : v;
        i = lane == dst ? ni : i;
    }
    // std::__adjust_heap(first, hole, len, value) followed by its __push_heap
    __device__ __forceinline__ void adjust(int hole, int len, float nv, int ni, int lane) {
        const int top = hole;
        int second = hole;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            if (val(second) < val(second - 1)) second--;
            move(hole, second, lane);
            hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            move(hole, second - 1, lane);
            hole = second - 1;
        }
        int parent = (hole - 1) / 2;
        while (hole > top && val(parent) < nv) {
            move(hole, parent, lane);
            hole = parent;
            parent = (hole - 1) / 2;
        }
        set(hole, nv, ni, lane);
    }
};

// libstdc++ on an array of (value, index) pairs in LDS, executed by one lane (M < 64 k: at most ~1100 elements)
struct SeqPairs {
    float *v;
    int *i;
    __device__ __forceinline__ void swap(int a, int b) {
        const float tv = v[a]; v[a] = v[b]; v[b] = tv;
        const int ti = i[a]; i[a] = i[b]; i[b] = ti;
    }
    __device__ void insertion_sort(int first, int last) {  // std::__insertion_sort
        if (first == last) return;
        for (int p = first + 1; p != last; ++p) {
            const float pv = v[p];
            const int pi = i[p];
            if (pv < v[first]) {
                for (int q = p; q != first; --q) { v[q] = v[q - 1]; i[q] = i[q - 1]; }
                v[first] = pv; i[first] = pi;
            } else {
                int pos = p, next = p - 1;
                while (pv < v[next]) { v[pos] = v[next]; i[pos] = i[next]; pos = next; --next; }
                v[pos] = pv; i[pos] = pi;
            }
        }
    }
    __device__ int partition_pivot(int first, int last) {  // std::__unguarded_partition_pivot
        const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
        if (v[a] < v[b]) {
            if (v[b] < v[c]) swap(first, b);
            else if (v[a] < v[c]) swap(first, c);
            else swap(first, a);
        } else if (v[a] < v[c]) swap(first, a);
        else if (v[b] < v[c]) swap(first, c);
        else swap(first, b);
        int lo = first + 1, hi = last;
        for (;;) {
            while (v[lo] < v[first]) ++lo;
            --hi;
            while (v[first] < v[hi]) --hi;
            if (!(lo < hi)) return lo;
            swap(lo, hi);
            ++lo;
        }
    }
    __device__ void adjust_heap(int first, int hole, int len, float nv, int ni) {
        const int top = hole;
        int second = hole;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            if (v[first + second] < v[first + second - 1]) second--;
            v[first + hole] = v[first + second]; i[first + hole] = i[first + second];
            hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            v[first + hole] = v[first + second - 1]; i[first + hole] = i[first + second - 1];
            hole = second - 1;
        }
        int parent = (hole - 1) / 2;
        while (hole > top && v[first + parent] < nv) {
            v[first + hole] = v[first + parent]; i[first + hole] = i[first + parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        v[first + hole] = nv; i[first + hole] = ni;
    }
    __device__ void heap_select(int first, int middle, int last) {  // std::__heap_select
        const int len = middle - first;
        if (len >= 2)
            for (int parent = (len - 2) / 2;; --parent) {
                adjust_heap(first, parent, len, v[first + parent], i[first + parent]);
                if (parent == 0) break;
            }
        for (int p = middle; p < last; ++p)
            if (v[p] < v[first]) {
                const float pv = v[p];
                const int pi = i[p];
                v[p] = v[first]; i[p] = i[first];
                adjust_heap(first, 0, len, pv, pi);
            }
    }
    __device__ void sort_heap(int first, int last) {  // std::__sort_heap
        while (last - first > 1) {
            --last;
            const float lv = v[last];
            const int li = i[last];
            v[last] = v[first]; i[last] = i[first];
            adjust_heap(first, 0, last - first, lv, li);
        }
    }
    // std::sort(first, last) for at most 63 elements: __introsort_loop (threshold 16; its recursion on the right part as
    // an explicit stack) then __final_insertion_sort.  k - 1 <= 16 never leaves the plain insertion sort.
    __device__ void sort(int first, int last) {
        if (first == last) return;
        int lg = 0;
        for (int t = last - first; t > 1; t >>= 1) ++lg;
        int stack_first[12], stack_last[12], stack_depth[12], sp = 0;
        stack_first[0] = first, stack_last[0] = last, stack_depth[0] = 2 * lg, sp = 1;
        while (sp > 0) {
            --sp;
            int f = stack_first[sp], l = stack_last[sp], depth = stack_depth[sp];
            while (l - f > 16) {
                if (depth == 0) {  // std::__partial_sort(f, l, l)
                    heap_select(f, l, l);
                    sort_heap(f, l);
                    break;
                }
                --depth;
                const int cut = partition_pivot(f, l);
                // the reference recurses into [cut, l) first and then loops on [f, cut); the two ranges are disjoint, so
                // the order in which they are finished does not change the result
                stack_first[sp] = cut, stack_last[sp] = l, stack_depth[sp] = depth, ++sp;
                l = cut;
            }
        }
        if (last - first > 16) {
            insertion_sort(first, first + 16);
            for (int p = first + 16; p != last; ++p) {  // __unguarded_insertion_sort
                const float pv = v[p];
                const int pi = i[p];
                int pos = p, next = p - 1;
                while (pv < v[next]) { v[pos] = v[next]; i[pos] = i[next]; pos = next; --next; }
                v[pos] = pv; i[pos] = pi;
            }
        } else {
            insertion_sort(first, last);
        }
    }
    __device__ void introselect(int first, int nth, int last, int depth_limit) {  // std::__introselect
        while (last - first > 3) {
            if (depth_limit == 0) {
                heap_select(first, nth + 1, last);
                swap(first, nth);
                return;
            }
            --depth_limit;
            const int cut = partition_pivot(first, last);
            if (cut <= nth) first = cut;
            else last = cut;
        }
        insertion_sort(first, last);
    }
};

// Several independent searches of one (B, D, k) in one launch: blockIdx.z picks the job.  The PointConv pyramid's five
// neighbour searches depend on the sampled coordinates only (pointconv.py:46 per level); launched together, the small
// levels fill the CUs the big one leaves idle instead of queueing behind it.
struct KnnJobs {
    rpe_knn_job job[RPE_KNN_MAX_JOBS];
};

// ---- k >= 2: cross-lane sorted list ----------------------------------------
template <int D, int QW, bool SMALL>  // SMALL: some job has M < 64 k (topk's nth_element form): LDS room for one row per wave
__global__ __launch_bounds__(kWavesPerBlock * RPE_WAVE) void knn_select_kernel(KnnJobs jobs, int k, int exact_ties, int row_stride) {
    extern __shared__ float seq_lds[];  // SMALL: per wave row_stride values then row_stride indices
    __shared__ float tile_ring[kWavesPerBlock][kRingSlots * 3 * RPE_WAVE];
    const rpe_knn_job &J = jobs.job[blockIdx.z];
    const float *__restrict__ inp = J.input;
    const float *__restrict__ qry = J.query;
    const int64_t in_sb = J.in_sb, in_sn = J.in_sn, in_sd = J.in_sd, q_sb = J.q_sb, q_sn = J.q_sn, q_sd = J.q_sd;
    const int M = J.M, Q = J.Q;
    int64_t *__restrict__ idx = J.idx;
    float *__restrict__ dist = J.dist;
    const int lane = rpe_lane();
    const int wave = rpe_uniform((int)(threadIdx.x >> 6));
    const int b = blockIdx.y;
    const int qbase = (blockIdx.x * kWavesPerBlock + wave) * QW;
    if (qbase >= Q) return;
    inp += (int64_t)b * in_sb;
    qry += (int64_t)b * q_sb;

    Queries<D, QW> qs;
    qs.load(qry, q_sn, q_sd, qbase, Q);

    float Ld[QW], tau[QW];
    int Li[QW];
#pragma unroll
    for (int j = 0; j < QW; ++j) {
        Ld[j] = INFINITY;
        Li[j] = 0;
        tau[j] = INFINITY;
    }
    // keep one entry more than asked for: it shows whether anything outside the top k ties with the k-th distance
    const int kk = (exact_ties && k < M && k < RPE_WAVE) ? k + 1 : k;

    // ---- pass 1: an upper bound for the kk-th smallest distance, so that the sweep below inserts ~kk candidates instead of
    // ~kk (1 + ln(M / kk)).  Every lane keeps the minimum over the points IT sees (one per tile): 64 distances of 64
    // distinct points, whose kk-th smallest therefore bounds the kk-th smallest of all M from above -- and, the best
    // few points mostly falling into different lanes, is about the (1.2 kk)-th smallest overall.  Points beyond the bound
    // cannot be among the kk best and never enter the serial insertion; equal to the bound they still compete (index order).
    if (M > 2 * RPE_WAVE) {
        float lm[QW];
#pragma unroll
        for (int j = 0; j < QW; ++j) lm[j] = INFINITY;
        TileStream<D> ts{inp, in_sn, in_sd, M, lane, tile_ring[wave]};
        ts.start();
        for (int base = 0; base < M; base += RPE_WAVE) {
            const bool valid = base + lane < M;
            float p[3];
            ts.next(base, p);
            const float pp = valid ? rpe_sqnorm<D>(p) : INFINITY;
#pragma unroll
            for (int j = 0; j < QW; ++j) {
                const float d = rpe_pair_dist<D>(qs.qm2[j], qs.qq[j], p, pp);
                lm[j] = d < lm[j] ? d : lm[j];  // (a NaN distance never lowers the minimum: the bound stays valid)
            }
        }
        ts.finish();
#pragma unroll
        for (int j = 0; j < QW; ++j) {
            int below = 0;  // how many of the 64 lane minima are strictly smaller than this lane's
#pragma unroll 16
            for (int l = 0; l < RPE_WAVE; ++l) below += rpe_readlane(lm[j], l) < lm[j] ? 1 : 0;
            float bound = below < kk ? lm[j] : -INFINITY;  // the kk-th smallest = the largest value with fewer than kk below it
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) bound = fmaxf(bound, __shfl_xor(bound, off));
            // strict comparisons below: step to the next float up so that d == bound still passes
            const int bits = __float_as_int(bound);
            tau[j] = !(bound < INFINITY) ? INFINITY : bound == 0.f ? __int_as_float(1) : bound > 0.f ? __int_as_float(bits + 1) : __int_as_float(bits - 1);
        }
    }
    float tau_cap[QW];
#pragma unroll
    for (int j = 0; j < QW; ++j) tau_cap[j] = tau[j];

    TileStream<D> sweep{inp, in_sn, in_sd, M, lane, tile_ring[wave]};
    sweep.start();
    for (int base = 0; base < M; base += RPE_WAVE) {
        const bool valid = base + lane < M;
        float p[3];
        sweep.next(base, p);
        const float pp = valid ? rpe_sqnorm<D>(p) : INFINITY;  // a lane past the end: |p|^2 = +inf -> d = +inf for every query
#pragma unroll
        for (int j = 0; j < QW; ++j) {
            const float d = rpe_pair_dist<D>(qs.qm2[j], qs.qq[j], p, pp);
            unsigned long long m = __ballot(d < tau[j]);
            while (m) {  // wave-uniform: candidates in index order
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const float nd = rpe_readlane(d, l);
                if (nd < tau[j]) {
                    const int ni = base + l;
                    // every listed entry has a smaller index than ni, so "nd < entry"
                    // (strict) keeps equal distances in index order
                    // lane r <- lane r-1 across the whole wave: DPP wave_shr:1 (a register move, ~10 cycles)
                    // instead of __shfl_up's ds_bpermute round trip; lane 0 keeps its own value
                    const float upd = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(Ld[j]), __float_as_int(Ld[j]), 0x138, 0xf, 0xf, false));
                    const int upi = __builtin_amdgcn_update_dpp(Li[j], Li[j], 0x138, 0xf, 0xf, false);
                    const bool gt = nd < Ld[j];
                    const bool gtp = (lane > 0) && (nd < upd);
                    Ld[j] = gt ? (gtp ? upd : nd) : Ld[j];
                    Li[j] = gt ? (gtp ? upi : ni) : Li[j];
                    tau[j] = fminf(rpe_readlane(Ld[j], kk - 1), tau_cap[j]);
                }
            }
        }
    }

    sweep.finish();

#pragma unroll
    for (int j = 0; j < QW; ++j) {
        const int qi = qbase + j;
        if (qi >= Q) continue;  // wave-uniform
        if (exact_ties && k < RPE_WAVE) {
            // equal neighbours among the kk best (lane r against lane r + 1)?  Then redo this query as libstdc++ would.
            const float nxt = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(Ld[j]), __float_as_int(Ld[j]), 0x130, 0xf, 0xf, false));  // wave_shl:1
            // exact_ties 1: only a tie ACROSS the boundary (k-th against (k+1)-th distance) can change WHICH neighbours are
            // returned; ties inside the top k change their order only and keep the sweep's index order.  3: any tie.
            unsigned long long dup = __ballot(lane + 1 < kk && Ld[j] == nxt);
            if (exact_ties == 1) dup &= 1ull << (k - 1);
            if (dup) {
                if (!SMALL || k * 64 <= M) {  // std::partial_sort: __heap_select over the row in index order, then __sort_heap
                    LaneHeap h;
                    h.v = INFINITY;
                    h.i = 0;
                    float top = 0.f;
                    for (int base = 0; base < M; base += RPE_WAVE) {
                        const int pi = base + lane;
                        const bool valid = pi < M;
                        float p[3] = {0.f, 0.f, 0.f};
                        if (valid) load_point<D>(inp, in_sn, in_sd, pi, p);
                        const float pp = valid ? rpe_sqnorm<D>(p) : INFINITY;
                        const float d = rpe_pair_dist<D>(qs.qm2[j], qs.qq[j], p, pp);
                        unsigned long long m;
                        if (base == 0) {  // the first k elements form the heap: std::__make_heap
                            h.v = d;
                            h.i = lane;
                            for (int parent = (k - 2) / 2;; --parent) {
                                h.adjust(parent, k, h.val(parent), rpe_readlane(h.i, parent), lane);
                                if (parent == 0) break;
                            }
                            top = h.val(0);
                            m = __ballot(lane >= k && d < top);
                        } else {
                            m = __ballot(d < top);
                        }
                        while (m) {  // __pop_heap(first, middle, i) for every later element below the heap's top, in order
                            const int l = __builtin_ctzll(m);
                            m &= m - 1;
                            const float nd = rpe_readlane(d, l);
                            if (nd < top) {
                                h.adjust(0, k, nd, base + l, lane);
                                top = h.val(0);
                            }
                        }
                    }
                    for (int last = k - 1; last >= 1; --last) {  // std::__sort_heap
                        const float lv = h.val(last);
                        const int li = rpe_readlane(h.i, last);
                        h.move(last, 0, lane);
                        h.adjust(0, last, lv, li, lane);
                    }
                    Ld[j] = h.v;
                    Li[j] = h.i;
                } else {  // std::nth_element(begin, begin + k - 1, end) + std::sort(begin, begin + k - 1) by one lane
                    SeqPairs sp{seq_lds + (size_t)wave * 2 * row_stride,
                                reinterpret_cast<int *>(seq_lds + (size_t)wave * 2 * row_stride + row_stride)};
                    for (int base = 0; base < M; base += RPE_WAVE) {
                        const int pi = base + lane;
                        if (pi < M) {
                            float p[3] = {0.f, 0.f, 0.f};
                            load_point<D>(inp, in_sn, in_sd, pi, p);
                            sp.v[pi] = rpe_pair_dist<D>(qs.qm2[j], qs.qq[j], p, rpe_sqnorm<D>(p));
                            sp.i[pi] = pi;
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) {
                        int lg = 0;
                        for (int t = M; t > 1; t >>= 1) ++lg;
                        sp.introselect(0, k - 1, M, 2 * lg);
                        sp.sort(0, k - 1);
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                    if (lane < k) {
                        Ld[j] = sp.v[lane];
                        Li[j] = sp.i[lane];
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (lane < k) {
            const int64_t o = ((int64_t)b * Q + qi) * k + lane;
            idx[o] = (int64_t)Li[j];
            if (dist) dist[o] = Ld[j];
        }
    }
}

// ---- k == 1: lane-local minimum ----------------------------------------------
template <int D, int QW>
__global__ __launch_bounds__(kWavesPerBlock * RPE_WAVE) void knn_nearest_kernel(KnnJobs jobs) {
    const rpe_knn_job &J = jobs.job[blockIdx.z];
    const float *__restrict__ inp = J.input;
    const float *__restrict__ qry = J.query;
    const int64_t in_sb = J.in_sb, in_sn = J.in_sn, in_sd = J.in_sd, q_sb = J.q_sb, q_sn = J.q_sn, q_sd = J.q_sd;
    const int M = J.M, Q = J.Q;
    int64_t *__restrict__ idx = J.idx;
    float *__restrict__ dist = J.dist;
    const int lane = rpe_lane();
    const int wave = rpe_uniform((int)(threadIdx.x >> 6));
    const int b = blockIdx.y;
    const int qbase = (blockIdx.x * kWavesPerBlock + wave) * QW;
    if (qbase >= Q) return;
    inp += (int64_t)b * in_sb;
    qry += (int64_t)b * q_sb;

    Queries<D, QW> qs;
    qs.load(qry, q_sn, q_sd, qbase, Q);

    float bd[QW];
    int bi[QW];
#pragma unroll
    for (int j = 0; j < QW; ++j) {
        bd[j] = INFINITY;
        bi[j] = 0x7fffffff;
    }

    __shared__ float tile_ring[kWavesPerBlock][kRingSlots * 3 * RPE_WAVE];
    TileStream<D> sweep{inp, in_sn, in_sd, M, lane, tile_ring[wave]};
    sweep.start();
    for (int base = 0; base < M; base += RPE_WAVE) {
        const int pi = base + lane;
        const bool valid = pi < M;
        float p[3];
        sweep.next(base, p);
        const float pp = valid ? rpe_sqnorm<D>(p) : INFINITY;  // past the end: d = +inf, never taken
#pragma unroll
        for (int j = 0; j < QW; ++j) {
            const float d = rpe_pair_dist<D>(qs.qm2[j], qs.qq[j], p, pp);
            const bool take = d < bd[j];  // strict: first index wins inside a lane
            bd[j] = take ? d : bd[j];
            bi[j] = take ? pi : bi[j];
        }
    }

    sweep.finish();
#pragma unroll
    for (int j = 0; j < QW; ++j) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(bd[j], off);
            const int oi = __shfl_xor(bi[j], off);
            const bool take = (od < bd[j]) || (od == bd[j] && oi < bi[j]);
            bd[j] = take ? od : bd[j];
            bi[j] = take ? oi : bi[j];
        }
        const int qi = qbase + j;
        if (qi < Q && lane == 0) {
            const int64_t o = (int64_t)b * Q + qi;
            idx[o] = bi[j] == 0x7fffffff ? 0 : (int64_t)bi[j];
            if (dist) dist[o] = bd[j];
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) void sqdist_kernel(const float *__restrict__ a, int64_t a_sb, int64_t a_sn, int64_t a_sd,
                                                     const float *__restrict__ p2, int64_t b_sb, int64_t b_sn, int64_t b_sd,
                                                     int N1, int N2, float *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    const int b = blockIdx.z;
    if (j >= N2) return;
    float q[3], p[3], qm2[3];
    load_point<D>(a + (int64_t)b * a_sb, a_sn, a_sd, i, q);
    load_point<D>(p2 + (int64_t)b * b_sb, b_sn, b_sd, j, p);
    const float qq = rpe_sqnorm<D>(q);
    const float pp = rpe_sqnorm<D>(p);
#pragma unroll
    for (int d = 0; d < 3; ++d) qm2[d] = -2.0f * q[d];
    out[((int64_t)b * N1 + i) * N2 + j] = rpe_pair_dist<D>(qm2, qq, p, pp);
}

int pick_qw(int B, int Q) {
    const long target = 2048;  // waves wanted in flight: 256 CUs x 4 SIMDs x 2 (1024 ... 4096 measure the same; 8192 and more are slower)
    for (int qw = 8; qw > 1; qw >>= 1)
        if ((long)B * ((Q + qw - 1) / qw) >= target) return qw;
    return 1;
}

template <int D, int QW>
int launch_knn(const KnnJobs &jobs, int njobs, int max_q, int min_m, int max_m, int B, int k, int ties, hipStream_t st) {
    const int per_block = kWavesPerBlock * QW;
    dim3 grid((max_q + per_block - 1) / per_block, B, njobs), block(kWavesPerBlock * RPE_WAVE);
    // k == 1 with M >= 64 is std::partial_sort with a one-element heap: the first minimum, which the lane-local kernel keeps
    if (k == 1 && (min_m >= 64 || !ties)) {
        hipLaunchKernelGGL((knn_nearest_kernel<D, QW>), grid, block, 0, st, jobs);
    } else if (ties && k < RPE_WAVE && min_m < 64 * k) {
        // some job is in topk's nth_element regime: every wave gets LDS room for one such row (values + indices)
        const int row = max_m < 64 * k ? max_m : 64 * k;  // only rows shorter than 64 k are ever copied
        const size_t lds = (size_t)kWavesPerBlock * 2 * row * sizeof(float);
        if (lds > 160 * 1024) return RPE_EUNSUPPORTED;
        if (lds > 64 * 1024) {  // beyond the default dynamic-LDS limit (k > 32 with a few thousand points)
            hipError_t e = hipFuncSetAttribute((const void *)knn_select_kernel<D, QW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((knn_select_kernel<D, QW, true>), grid, block, lds, st, jobs, k, ties, row);
    } else {
        hipLaunchKernelGGL((knn_select_kernel<D, QW, false>), grid, block, 0, st, jobs, k, ties, 0);
    }
    return rpe_launch_status();
}

template <int D>
int launch_knn_d(int qw, const KnnJobs &jobs, int njobs, int max_q, int min_m, int max_m, int B, int k, int ties, hipStream_t st) {
    switch (qw) {
        case 1: return launch_knn<D, 1>(jobs, njobs, max_q, min_m, max_m, B, k, ties, st);
        case 2: return launch_knn<D, 2>(jobs, njobs, max_q, min_m, max_m, B, k, ties, st);
        case 4: return launch_knn<D, 4>(jobs, njobs, max_q, min_m, max_m, B, k, ties, st);
        default: return launch_knn<D, 8>(jobs, njobs, max_q, min_m, max_m, B, k, ties, st);
    }
}

}  // namespace

RPE_API int rpe_knn_multi(const rpe_knn_job *jobs, int njobs, int B, int D, int k, int tie_mode, rpe_stream_t stream) {
    if (!jobs || njobs < 1 || njobs > RPE_KNN_MAX_JOBS || B < 0 || D < 1 || D > 3 || k < 1) return RPE_EINVAL;
    if (tie_mode != RPE_KNN_TIES_TORCH && tie_mode != RPE_KNN_TIES_SET && tie_mode != RPE_KNN_TIES_INDEX) return RPE_EINVAL;
    if (k > RPE_WAVE) return RPE_EUNSUPPORTED;
    if (B > 65535) return RPE_EUNSUPPORTED;
    KnnJobs packed;
    int max_q = 0, min_m = 0x7fffffff, max_m = 0;
    long total_q = 0;
    for (int i = 0; i < njobs; ++i) {
        const rpe_knn_job &j = jobs[i];
        if (!j.input || !j.query || !j.idx || j.M <= 0 || j.Q < 0 || k > j.M) return RPE_EINVAL;
        packed.job[i] = j;
        max_q = j.Q > max_q ? j.Q : max_q;
        min_m = j.M < min_m ? j.M : min_m;
        max_m = j.M > max_m ? j.M : max_m;
        total_q += j.Q;
    }
    if (B == 0 || max_q == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int qw = pick_qw(B, (int)total_q);
    if (D == 3) return launch_knn_d<3>(qw, packed, njobs, max_q, min_m, max_m, B, k, tie_mode, st);
    if (D == 2) return launch_knn_d<2>(qw, packed, njobs, max_q, min_m, max_m, B, k, tie_mode, st);
    return launch_knn_d<1>(qw, packed, njobs, max_q, min_m, max_m, B, k, tie_mode, st);
}

RPE_API int rpe_knn(const float *input, int64_t in_sb, int64_t in_sn, int64_t in_sd, const float *query, int64_t q_sb,
                    int64_t q_sn, int64_t q_sd, int B, int M, int Q, int D, int k, int tie_mode, int64_t *idx, float *dist,
                    rpe_stream_t stream) {
    if (!input || !query || !idx || B < 0 || M <= 0 || Q < 0 || D < 1 || D > 3) return RPE_EINVAL;
    if (k < 1 || k > M) return RPE_EINVAL;
    const rpe_knn_job job{input, in_sb, in_sn, in_sd, query, q_sb, q_sn, q_sd, M, Q, idx, dist};
    return rpe_knn_multi(&job, 1, B, D, k, tie_mode, stream);
}

RPE_API int rpe_squared_distance(const float *xyz1, int64_t a_sb, int64_t a_sn, int64_t a_sd, const float *xyz2,
                                 int64_t b_sb, int64_t b_sn, int64_t b_sd, int B, int N1, int N2, int D, float *out,
                                 rpe_stream_t stream) {
    if (!xyz1 || !xyz2 || !out || B < 0 || N1 < 0 || N2 < 0 || D < 1 || D > 3) return RPE_EINVAL;
    if (B > 65535 || N1 > 65535) return RPE_EUNSUPPORTED;
    if (B == 0 || N1 == 0 || N2 == 0) return 0;
    dim3 grid((N2 + 255) / 256, N1, B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (D == 3)
        hipLaunchKernelGGL(sqdist_kernel<3>, grid, block, 0, st, xyz1, a_sb, a_sn, a_sd, xyz2, b_sb, b_sn, b_sd, N1, N2, out);
    else if (D == 2)
        hipLaunchKernelGGL(sqdist_kernel<2>, grid, block, 0, st, xyz1, a_sb, a_sn, a_sd, xyz2, b_sb, b_sn, b_sd, N1, N2, out);
    else
        hipLaunchKernelGGL(sqdist_kernel<1>, grid, block, 0, st, xyz1, a_sb, a_sn, a_sd, xyz2, b_sb, b_sn, b_sd, N1, N2, out);
    return rpe_launch_status();
}
