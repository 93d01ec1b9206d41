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

#include <type_traits>

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

// __adjust_heap(first, node, len, value) + its __push_heap for the lane-distributed heap of a compile-time k -- the step of
// __make_heap, __heap_select (~100 replacements per replayed row) and __sort_heap -- as a decision tree over CONSTANT lanes: the children of node n sit in lanes
// 2n + 1 and 2n + 2, so every v_readlane / v_writelane has an immediate lane select and every comparison is a scalar integer
// compare on order-preserving keys (sign bit flipped, negative values complemented; a distance is never -0).  libstdc++ walks
// the hole down to a leaf along the larger children (the right one unless it is smaller than the left) and pushes the value
// back up while the parent is smaller; the values along that path do not increase, so the value ends at the first path node
// whose path child is smaller than it and every node below keeps its old value: walking down and stopping there gives the
// same array.  The generic LaneHeap::adjust (dynamic lane selects, float compares, masked moves) costs ~1100 cycles a
// replacement for a lone wave.
__device__ __forceinline__ unsigned heap_key(float d) {
    const unsigned b = (unsigned)__float_as_int(d);
    return b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float heap_dist(unsigned k) { return __int_as_float((int)((k & 0x80000000u) ? k ^ 0x80000000u : ~k)); }
template <int LANE>
__device__ __forceinline__ void heap_put(unsigned &kv, int &iv, unsigned nk, int ni) {  // (wave-uniform nk, ni)
    asm("v_writelane_b32 %0, %1, %2" : "+v"(kv) : "s"(nk), "n"(LANE));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(iv) : "s"(ni), "n"(LANE));
}
template <int NODE, int LEN>
__device__ __forceinline__ void heap_sift(unsigned &kv, int &iv, unsigned nk, int ni) {  // __adjust_heap(first, NODE, LEN, value)
    constexpr int L = 2 * NODE + 1, R = L + 1;
    if constexpr (L >= LEN) {
        heap_put<NODE>(kv, iv, nk, ni);
    } else if constexpr (R >= LEN) {  // LEN even: the last inner node has a left child only, and that child is a leaf
        const unsigned kl = (unsigned)__builtin_amdgcn_readlane((int)kv, L);
        if (kl < nk) {
            heap_put<NODE>(kv, iv, nk, ni);
        } else {
            heap_put<NODE>(kv, iv, kl, __builtin_amdgcn_readlane(iv, L));
            heap_put<L>(kv, iv, nk, ni);
        }
    } else {
        const unsigned kl = (unsigned)__builtin_amdgcn_readlane((int)kv, L), kr = (unsigned)__builtin_amdgcn_readlane((int)kv, R);
        if (kr < kl) {
            if (kl < nk) {
                heap_put<NODE>(kv, iv, nk, ni);
            } else {
                heap_put<NODE>(kv, iv, kl, __builtin_amdgcn_readlane(iv, L));
                heap_sift<L, LEN>(kv, iv, nk, ni);
            }
        } else {
            if (kr < nk) {
                heap_put<NODE>(kv, iv, nk, ni);
            } else {
                heap_put<NODE>(kv, iv, kr, __builtin_amdgcn_readlane(iv, R));
                heap_sift<R, LEN>(kv, iv, nk, ni);
            }
        }
    }
}
template <int K, int PARENT>
__device__ __forceinline__ void heap_make(unsigned &kv, int &iv) {  // std::__make_heap: parents (K - 2) / 2 .. 0
    heap_sift<PARENT, K>(kv, iv, (unsigned)__builtin_amdgcn_readlane((int)kv, PARENT), __builtin_amdgcn_readlane(iv, PARENT));
    if constexpr (PARENT > 0) heap_make<K, PARENT - 1>(kv, iv);
}
template <int LAST>
__device__ __forceinline__ void heap_sort(unsigned &kv, int &iv) {  // std::__sort_heap: last = K - 1 .. 1
    const unsigned lv = (unsigned)__builtin_amdgcn_readlane((int)kv, LAST);
    const int li = __builtin_amdgcn_readlane(iv, LAST);
    heap_put<LAST>(kv, iv, (unsigned)__builtin_amdgcn_readlane((int)kv, 0), __builtin_amdgcn_readlane(iv, 0));
    heap_sift<0, LAST>(kv, iv, lv, li);
    if constexpr (LAST > 1) heap_sort<LAST - 1>(kv, iv);
}

// std::partial_sort of one query's row for a compile-time k on that tree: __make_heap of the first K distances, every later
// one below the top replaces it (__heap_select), __sort_heap; the result in lane r = rank r.
template <int D, int K>
__device__ __forceinline__ void heap_replay_static(float &Ld, int &Li, const float (&qm2)[3], float qq, const float *__restrict__ inp,
                                                   int64_t in_sn, int64_t in_sd, int M, int lane) {
    unsigned kv = heap_key(INFINITY);
    int iv = 0;
    float top = INFINITY;
    constexpr int kTieTiles = 4;  // (eight tiles a round make the kernel spill and slow its sweeps: 125 -> 138 us)
    for (int base0 = 0; base0 < M; base0 += kTieTiles * RPE_WAVE) {
        float d[kTieTiles];
        unsigned long long m[kTieTiles];
        float pt[kTieTiles][3];
#pragma unroll
        for (int u = 0; u < kTieTiles; ++u) load_point<D>(inp, in_sn, in_sd, min(base0 + u * RPE_WAVE + lane, M - 1), pt[u]);
#pragma unroll
        for (int u = 0; u < kTieTiles; ++u) {
            const float pp = base0 + u * RPE_WAVE + lane < M ? rpe_sqnorm<D>(pt[u]) : INFINITY;
            d[u] = rpe_pair_dist<D>(qm2, qq, pt[u], pp);
        }
        if (base0 == 0) {
            kv = heap_key(d[0]);
            iv = lane;
            if constexpr (K >= 2) heap_make<K, (K - 2) / 2>(kv, iv);
            top = heap_dist((unsigned)__builtin_amdgcn_readlane((int)kv, 0));
        }
        unsigned long long any = 0ull;
#pragma unroll
        for (int u = 0; u < kTieTiles; ++u) {
            m[u] = __ballot((base0 > 0 || u > 0 || lane >= K) && d[u] < top);  // (top only falls: a superset of what passes later)
            any |= m[u];
        }
        if (!any) continue;
#pragma unroll
        for (int u = 0; u < kTieTiles; ++u) {
            unsigned long long mu = m[u] & __ballot(d[u] < top);  // against the top as it is now
            while (mu) {  // __pop_heap(first, middle, i) for every later element below the heap's top, in order
                const int l = __builtin_ctzll(mu);
                heap_sift<0, K>(kv, iv, heap_key(rpe_readlane(d[u], l)), base0 + u * RPE_WAVE + l);
                top = heap_dist((unsigned)__builtin_amdgcn_readlane((int)kv, 0));
                mu &= (~1ull << l) & __ballot(d[u] < top);  // what is left of this tile, against the new top
            }
        }
    }
    if constexpr (K >= 2) heap_sort<K - 1>(kv, iv);
    Ld = heap_dist(kv);
    Li = iv;
}

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
    __device__ void median_to_first(int first, int last) {  // std::__move_median_to_first(first, first + 1, mid, last - 1)
        const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
        if (v[a] < v[b]) {
            if (v[b] < v[c]) swap(first, b);
            else if (v[a] < v[c]) swap(first, c);
            else swap(first, a);
        } else if (v[a] < v[c]) swap(first, a);
        else if (v[b] < v[c]) swap(first, c);
        else swap(first, b);
    }
    __device__ int partition_pivot(int first, int last) {  // std::__unguarded_partition_pivot
        median_to_first(first, last);
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

// ---- the same libstdc++ code paths, executed by the whole wave ------------------------------------------------------------
// One lane walking an LDS array pays ~100 cycles per element (80 us for nth_element over 512 distances: the tail of every
// small-level search that has a tie anywhere).  Hoare's partition is a deterministic pairing, though: with
// A = positions of [lo, hi) whose value is not below the pivot (where the upward scan stops) and B = positions whose value is
// not above it, plus the pivot's own slot as the last stop (where the downward scan stops), the sequential loop swaps the
// t-th element of A counted from the left with the t-th element of B counted from the right for t = 0 .. T - 1, T = the
// first t whose pair has crossed, and returns the T-th element of A or the upper slot of the last swap, whichever comes
// first.  (Until they cross, the scans stay inside the window between the last swapped pair, so membership can be taken
// from the values before any swap.)  Ranks come from ballots row by row; positions go to two
// 16-bit LDS arrays; all swaps of a partition happen at once.
struct WavePartition {
    SeqPairs sp;
    unsigned short *posA, *posB;  // [row_stride + 1] each
    int lane;

    __device__ __forceinline__ void sync() const {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    // std::__unguarded_partition_pivot(first, last)
    __device__ int partition_pivot(int first, int last) {
        if (lane == 0) sp.median_to_first(first, last);
        sync();
        const float pivot = sp.v[first];
        const int lo0 = first + 1, hi0 = last, rows = (hi0 - lo0 + RPE_WAVE - 1) / RPE_WAVE;
        int nA = 0, nB = 0;
        for (int j = 0; j < rows; ++j) {  // A ascending
            const int pos = lo0 + j * RPE_WAVE + lane;
            const bool in = pos < hi0 && !(sp.v[min(pos, hi0 - 1)] < pivot);
            const unsigned long long m = __ballot(in);
            if (in) posA[nA + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (unsigned short)pos;
            nA += (int)__builtin_popcountll(m);
        }
        for (int j = rows - 1; j >= 0; --j) {  // B descending
            const int pos = lo0 + j * RPE_WAVE + lane;
            const bool in = pos < hi0 && !(pivot < sp.v[min(pos, hi0 - 1)]);
            const unsigned long long m = __ballot(in);
            const unsigned long long above = lane == RPE_WAVE - 1 ? 0ull : m >> (lane + 1);
            if (in) posB[nB + (int)__builtin_popcountll(above)] = (unsigned short)pos;
            nB += (int)__builtin_popcountll(m);
        }
        if (lane == 0) posB[nB] = (unsigned short)first;  // the pivot's own slot stops the downward scan
        ++nB;
        sync();
        const int lim = min(nA, nB);
        int T = 0;  // pairs that have not crossed (the condition is monotone in t)
        for (int t0 = 0; t0 < lim; t0 += RPE_WAVE) {
            const int t = t0 + lane;
            const bool ok = t < lim && posA[min(t, lim - 1)] < posB[min(t, lim - 1)];
            const int c = (int)__builtin_popcountll(__ballot(ok));
            T += c;
            if (c < RPE_WAVE) break;
        }
        // the upward scan after swap T - 1 stops at the next element of A -- or, if there is none before it, at the slot
        // that swap just filled with a value from A
        int cut = T < nA ? (int)posA[T] : hi0;
        if (T > 0) cut = min(cut, (int)posB[T - 1]);
        for (int t0 = 0; t0 < T; t0 += RPE_WAVE) {
            const int t = t0 + lane;
            if (t < T) sp.swap((int)posA[t], (int)posB[t]);
        }
        sync();
        return cut;
    }
    // std::__introselect(first, nth, last, depth_limit)
    __device__ void introselect(int first, int nth, int last, int depth_limit) {
        while (last - first > 3) {
            if (depth_limit == 0) {
                if (lane == 0) {
                    sp.heap_select(first, nth + 1, last);
                    sp.swap(first, nth);
                }
                sync();
                return;
            }
            --depth_limit;
            const int cut = partition_pivot(first, last);
            if (cut <= nth) first = cut;
            else last = cut;
        }
        if (lane == 0) sp.insertion_sort(first, last);
        sync();
    }
};

// ---- what happens to a finished (k + 1)-entry list: the tie check, the libstdc++ restatement if it fires, the store ----
// One query redone the libstdc++ way (wave-uniform coordinates qm2 = -2 q, qq = |q|^2): the result in lane r = rank r.
template <int D, bool SMALL>
__device__ __forceinline__ void resolve_ties(float &Ld, int &Li, const float (&qm2)[3], float qq, const float *__restrict__ inp,
                                             int64_t in_sn, int64_t in_sd, int M, int k, int lane, int wave, float *seq_lds, int row_stride,
                                             int wide) {
    if ((!SMALL || k * 64 <= M) && k == 16) {  // (the k of the PointConv / Correlation3D searches)
        heap_replay_static<D, 16>(Ld, Li, qm2, qq, inp, in_sn, in_sd, M, lane);
    } else if ((!SMALL || k * 64 <= M) && k == 3) {  // (knn_interpolation, backwarp_3d)
        heap_replay_static<D, 3>(Ld, Li, qm2, qq, inp, in_sn, in_sd, M, lane);
    } else if (!SMALL || k * 64 <= M) {  // std::partial_sort: __heap_select over the row in index order, then __sort_heap
        LaneHeap h;
        h.v = INFINITY;
        h.i = 0;
        float top = INFINITY;
        // Four tiles a round: this wave is alone with its row (a lone wave issues a dependent instruction every
        // ~10 cycles and paid ~800 per 64-point tile, most rounds without a single candidate), so the loads and the
        // distance arithmetic of 256 points run as four independent chains and one test skips the round.
        constexpr int kTieTiles = 4;
        for (int base0 = 0; base0 < M; base0 += kTieTiles * RPE_WAVE) {
            float d[kTieTiles];
            unsigned long long m[kTieTiles];
            float pt[kTieTiles][3];
#pragma unroll
            for (int u = 0; u < kTieTiles; ++u) load_point<D>(inp, in_sn, in_sd, min(base0 + u * RPE_WAVE + lane, M - 1), pt[u]);
#pragma unroll
            for (int u = 0; u < kTieTiles; ++u) {
                const float pp = base0 + u * RPE_WAVE + lane < M ? rpe_sqnorm<D>(pt[u]) : INFINITY;
                d[u] = rpe_pair_dist<D>(qm2, qq, pt[u], pp);
            }
            if (base0 == 0) {  // the first k elements form the heap: std::__make_heap
                h.v = d[0];
                h.i = lane;
                for (int parent = (k - 2) / 2;; --parent) {
                    h.adjust(parent, k, h.val(parent), rpe_readlane(h.i, parent), lane);
                    if (parent == 0) break;
                }
                top = h.val(0);
            }
            unsigned long long any = 0ull;
#pragma unroll
            for (int u = 0; u < kTieTiles; ++u) {
                m[u] = __ballot((base0 > 0 || u > 0 || lane >= k) && d[u] < top);  // (top only falls: a superset of what passes later)
                any |= m[u];
            }
            if (!any) continue;
#pragma unroll
            for (int u = 0; u < kTieTiles; ++u) {
                unsigned long long mu = m[u] & __ballot(d[u] < top);  // against the top as it is now
                while (mu) {  // __pop_heap(first, middle, i) for every later element below the heap's top, in order
                    const int l = __builtin_ctzll(mu);
                    h.adjust(0, k, rpe_readlane(d[u], l), base0 + u * RPE_WAVE + l, lane);
                    top = h.val(0);
                    mu &= (~1ull << l) & __ballot(d[u] < top);  // what is left of this tile, against the new top
                }
            }
        }
        for (int last = k - 1; last >= 1; --last) {  // std::__sort_heap
            const float lv = h.val(last);
            const int li = rpe_readlane(h.i, last);
            h.move(last, 0, lane);
            h.adjust(0, last, lv, li, lane);
        }
        Ld = h.v;
        Li = h.i;
    } else {  // std::nth_element(begin, begin + k - 1, end) + std::sort(begin, begin + k - 1) by one lane
        SeqPairs sp{seq_lds + (size_t)wave * 2 * row_stride,
                    reinterpret_cast<int *>(seq_lds + (size_t)wave * 2 * row_stride + row_stride)};
        for (int base = 0; base < M; base += RPE_WAVE) {
            const int pi = base + lane;
            if (pi < M) {
                float p[3] = {0.f, 0.f, 0.f};
                load_point<D>(inp, in_sn, in_sd, pi, p);
                sp.v[pi] = rpe_pair_dist<D>(qm2, qq, p, rpe_sqnorm<D>(p));
                sp.i[pi] = pi;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        int lg = 0;
        for (int t = M; t > 1; t >>= 1) ++lg;
        if (wide) {
            unsigned short *pos = reinterpret_cast<unsigned short *>(seq_lds + (size_t)kWavesPerBlock * 2 * row_stride) + (size_t)wave * 2 * (row_stride + 2);
            WavePartition wp{sp, pos, pos + row_stride + 2, lane};
            wp.introselect(0, k - 1, M, 2 * lg);
        } else {  // (no LDS left for the position arrays: rows of several thousand distances, k > 32)
            if (lane == 0) sp.introselect(0, k - 1, M, 2 * lg);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
        if (k - 1 <= 16) {
            // std::sort of at most 16 elements is __insertion_sort alone, and that is a STABLE sort: rank by (value, position)
            const bool mine = lane < k - 1;
            const float v0 = lane < k ? sp.v[lane] : INFINITY;
            const int i0 = lane < k ? sp.i[lane] : 0;
            int rank = 0;
            for (int t = 0; t < k - 1; ++t) {
                const float vt = rpe_readlane(v0, t);
                rank += (vt < v0 || (vt == v0 && t < lane)) ? 1 : 0;
            }
            rank = mine ? rank : lane;  // the k-th element (and the idle lanes) stay where they are
            Ld = __int_as_float(__builtin_amdgcn_ds_permute(rank << 2, __float_as_int(v0)));
            Li = __builtin_amdgcn_ds_permute(rank << 2, i0);
        } else {
            if (lane == 0) sp.sort(0, k - 1);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (lane < k) {
                Ld = sp.v[lane];
                Li = sp.i[lane];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// equal neighbours among the kk best (lane r against lane r + 1)?  exact_ties 1: only a tie ACROSS the boundary (k-th against
// (k+1)-th distance) can change WHICH neighbours are returned; ties inside the top k change their order only and keep the
// sweep's index order.  3: any tie.
__device__ __forceinline__ bool has_ties(float Ld, int k, int kk, int exact_ties, int lane) {
    if (!exact_ties || k >= RPE_WAVE) return false;
    const float nxt = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(Ld), __float_as_int(Ld), 0x130, 0xf, 0xf, false));  // wave_shl:1
    unsigned long long dup = __ballot(lane + 1 < kk && Ld == nxt);
    if (exact_ties == 1) dup &= 1ull << (k - 1);
    return dup != 0ull;
}

// Ld / Li: lane r holds rank r of the query (wave-uniform coordinates qm2 = -2 q, qq = |q|^2).
template <int D, bool SMALL>
__device__ __forceinline__ void finish_query(float Ld, int Li, const float (&qm2)[3], float qq, const float *__restrict__ inp,
                                             int64_t in_sn, int64_t in_sd, int M, int k, int kk, int exact_ties, int lane, int wave,
                                             float *seq_lds, int row_stride, int wide, int64_t out_row, int64_t *__restrict__ idx,
                                             float *__restrict__ dist) {
    if (has_ties(Ld, k, kk, exact_ties, lane))  // then redo this query as libstdc++ would
        resolve_ties<D, SMALL>(Ld, Li, qm2, qq, inp, in_sn, in_sd, M, k, lane, wave, seq_lds, row_stride, wide);
    if (lane < k) {
        const int64_t o = out_row * k + lane;
        idx[o] = (int64_t)Li;
        if (dist) dist[o] = Ld;
    }
}

// The matrix kernel's waves own 16 queries each and ~1.4 % of the queries of a real cloud tie (|q|^2 + |p|^2 - 2 q.p cancels
// to a few thousand distinct values among the nearest neighbours); one takes a wave 30-50 us, and the wave with the most of
// them set the kernel's time.  There a tied query goes to a queue of the workgroup, and once every wave is through its own
// queries the four waves take the queued ones in turn.  (With at most 8 queries a wave the insertion kernel measured
// faster redoing them on the spot.)
struct TieQueue {
    int n;
    int query[kWavesPerBlock * 16];
};

template <int D>
__device__ __forceinline__ void drain_ties(TieQueue *tq, const rpe_knn_job &J, int b, int k, int lane, int wave) {
    __syncthreads();
    const int n = tq->n;
    const float *inp = J.input + (int64_t)b * J.in_sb;
    for (int t = wave; t < n; t += kWavesPerBlock) {
        const int qi = tq->query[t];
        float q[3], qm2[3];
        load_point<D>(J.query + (int64_t)b * J.q_sb, J.q_sn, J.q_sd, qi, q);
#pragma unroll
        for (int d = 0; d < 3; ++d) q[d] = rpe_uniform(q[d]);
        const float qq = rpe_sqnorm<D>(q);
#pragma unroll
        for (int d = 0; d < 3; ++d) qm2[d] = -2.0f * q[d];
        float Ld = 0.f;
        int Li = 0;
        resolve_ties<D, false>(Ld, Li, qm2, qq, inp, J.in_sn, J.in_sd, J.M, k, lane, wave, nullptr, 0, 0);
        if (lane < k) {
            const int64_t o = ((int64_t)b * J.Q + qi) * k + lane;
            J.idx[o] = (int64_t)Li;
            if (J.dist) J.dist[o] = Ld;
        }
    }
}

// Several independent searches of one (B, D, k) in one launch: blockIdx.z picks the job.  The PointConv pyramid's five
// neighbour searches depend on the sampled coordinates only (pointconv.py:46 per level); launched together, the small
// levels fill the CUs the big one leaves idle instead of queueing behind it.
struct KnnJobs {
    rpe_knn_job job[RPE_KNN_MAX_JOBS];
};

// ---- k >= 2: cross-lane sorted list ----------------------------------------
template <int D, int QW, bool SMALL>  // SMALL: some job has M < 64 k (topk's nth_element form): LDS room for one row per wave
__global__ __launch_bounds__(kWavesPerBlock * RPE_WAVE) void knn_select_kernel(KnnJobs jobs, int k, int exact_ties, int row_stride, int wide) {
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
        finish_query<D, SMALL>(Ld[j], Li[j], qs.qm2[j], qs.qq[j], inp, in_sn, in_sd, M, k, kk, exact_ties, lane, wave, seq_lds, row_stride, wide,
                               (int64_t)b * Q + qi, idx, dist);
    }
}

// ---- k >= 2 on large clouds: the distances on the matrix pipe -----------------------------------------------------------
// squared_distance IS a matrix product (wrapper.py:49: -2 * matmul(xyz1, xyz2^T), then + |xyz1|^2, then + |xyz2|^2), and
// v_mfma_f32_16x16x4_f32 accumulates its four k-slots as a sequential fp32 fma chain from C (measured bit for bit against
// fmaf on this chip), so with A[p][0..3] = (px, py, pz, 1) and B[0..3][q] = (-2 qx, -2 qy, -2 qz, |q|^2) one instruction
// yields fl(fl(-2 q.p) + |q|^2) for 16 points x 16 queries exactly as rpe_pair_dist computes it; the four-block
// v_mfma_f32_16x16x1_f32 with A = |p|^2, B = 1 then adds |p|^2 to four such tiles (64 points) at once.  5 matrix
// instructions replace 7 VALU instructions per 64 pairs x 16.
//
// A wave owns 16 queries; the block's four waves share the cloud, which streams through an LDS ring (LDS-DMA, counted
// waits).  D layout: lane (g = l >> 4, c = l & 15) works for query c; register 4 b + r holds point 16 b + 4 g + r of the
// 64-point step.  A lane therefore sees one query and a fixed quarter of the cloud, and the selection
// needs no cross-lane traffic in the sweeps -- threshold + collect instead of sorted insertion:
//   pass A  running minimum per register: for every query 64 minima over 64 DISJOINT subsets of the cloud (4 lanes x 16
//           registers); their kk-th smallest bounds the kk-th smallest distance from above (about the 1.2 kk-th smallest
//           in practice).  Found by a bitonic sort of the 64 values where they are;
//   pass B  same sweep; every lane appends the points below the bound to its own short list in LDS, branch-free: the entry
//           is always written, the list only grows on a hit.  ~1.2 kk entries per query over its four lanes;
//   then    per query: the entries of its four lists are ranked by (distance, index) with a counting loop, permuted into
//           the "lane r = rank r" form and handed to finish_query (tie check, libstdc++ restatement, store) exactly as the
//           insertion kernel does.  A lane list that fills up (15 entries: heavy ties) sends the query to the serial sweep.
typedef float knn_f32x4 __attribute__((ext_vector_type(4)));
typedef float knn_f32x8 __attribute__((ext_vector_type(8)));
typedef float knn_f32x16 __attribute__((ext_vector_type(16)));
// v_min_f32 as it is (fminf adds a canonicalising v_max per operand; NaN distances are not modelled)
__device__ __forceinline__ float knn_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float knn_min3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
constexpr int kMq = 16;                   // queries per wave
constexpr int kRow = 80;                  // floats per coordinate row of a ring slot: 64 + 16, so the four k-slots of an A fragment hit different banks
constexpr int kSlotFloats = 3 * kRow;
constexpr int kLaneList = 16;             // entries per lane list (15 usable: the slot after the last entry is scratch)
#ifndef RPE_KNN_MATRIX_MIN_M
#define RPE_KNN_MATRIX_MIN_M 1024
#endif
constexpr int kMatrixMinM = RPE_KNN_MATRIX_MIN_M;  // below this the fixed costs (bound sort, ranking) outweigh the sweep
constexpr long kMatrixMinQueries = 16384;  // B * Q: 16 queries a wave; below one wave per SIMD the insertion kernel is faster (2048^2, B = 4: 32 vs 38 us)
constexpr int kChunkSteps = 4;            // a chunk = 4 steps of 64 points; wave w of the block loads step w of every chunk
#ifndef RPE_KNN_CHUNKS
#define RPE_KNN_CHUNKS 4
#define RPE_KNN_AHEAD 2
#endif
constexpr int kChunks = RPE_KNN_CHUNKS;      // ring depth in chunks (a power of two)
constexpr int kChunksAhead = RPE_KNN_AHEAD;  // chunk t + 2 is requested before chunk t is used (kChunks >= kChunksAhead + 2)

struct MfmaRing {
    float ring[kChunks][kChunkSteps][kSlotFloats];  // the cloud streams through here once per pass, shared by the block's four waves
    float ones[kRow];
};
struct MfmaBlockLds {
    MfmaRing r;
    int count[kWavesPerBlock][RPE_WAVE];
    unsigned long long list[kWavesPerBlock][kLaneList][RPE_WAVE];  // entry s of lane l at [s][l] (lane-interleaved: no bank conflicts): (index << 32) | distance bits
};

// one query, the insertion sweep with plain loads: the fallback of the matrix kernel when a lane list fills up.  Rare (a lane
// holds ~5 entries on average, 15 fit: a few queries per launch) but the wave that meets one runs on alone after every other
// wave has finished, so its latency is the launch's: eight tiles a round (24 loads in flight, one "anything below the
// bound?" test) instead of one tile per load-wait-compare round: 128 rounds of ~0.16 us for a cloud of 8192 were +20 us on
// a 117 us launch (tools/experiments/knn_k_cliff.py).
template <int D>
__device__ void serial_select(const float *__restrict__ inp, int64_t in_sn, int64_t in_sd, int M, const float (&qm2)[3], float qq,
                              int kk, int lane, float &Ld, int &Li) {
    Ld = INFINITY;
    Li = 0;
    float tau = INFINITY;
    constexpr int kTiles = 8;
    for (int base0 = 0; base0 < M; base0 += kTiles * RPE_WAVE) {
        float pt[kTiles][3], d[kTiles];
#pragma unroll
        for (int u = 0; u < kTiles; ++u) load_point<D>(inp, in_sn, in_sd, min(base0 + u * RPE_WAVE + lane, M - 1), pt[u]);
        unsigned long long any = 0ull;
#pragma unroll
        for (int u = 0; u < kTiles; ++u) {
            const float pp = base0 + u * RPE_WAVE + lane < M ? rpe_sqnorm<D>(pt[u]) : INFINITY;
            d[u] = rpe_pair_dist<D>(qm2, qq, pt[u], pp);
            any |= __ballot(d[u] < tau);  // (tau only falls: a superset of what passes below)
        }
        if (!any) continue;
#pragma unroll
        for (int u = 0; u < kTiles; ++u) {
            unsigned long long m = __ballot(d[u] < tau);
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const float nd = rpe_readlane(d[u], l);
                if (nd < tau) {
                    const int ni = base0 + u * RPE_WAVE + l;
                    const float upd = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(Ld), __float_as_int(Ld), 0x138, 0xf, 0xf, false));
                    const int upi = __builtin_amdgcn_update_dpp(Li, Li, 0x138, 0xf, 0xf, false);
                    const bool gt = nd < Ld;
                    const bool gtp = (lane > 0) && (nd < upd);
                    Ld = gt ? (gtp ? upd : nd) : Ld;
                    Li = gt ? (gtp ? upi : ni) : Li;
                    tau = rpe_readlane(Ld, kk - 1);
                }
            }
        }
    }
}

template <int D>
struct MfmaSweep {
    const float *inp;
    int64_t sn, sd;
    int M, lane, wave;
    MfmaRing *L;
    float qb;            // B fragment (queries): lane (kq = l >> 4, col = l & 15): -2 q[kq] of query col, |q|^2 for kq = 3
    int aoff;            // A fragment (points) source of this lane: float offset inside a step's slot, or -1: the row of ones

    // LDS-DMA, no staging registers; every request issues exactly D loads (lanes / steps past the end re-read the last
    // point), so the counted waits are exact
    __device__ __forceinline__ void request(int chunk) const {
        const int base = (chunk * kChunkSteps + wave) * RPE_WAVE;
        const int pi = min(base + lane, M - 1);
        const float *src = inp + (int64_t)pi * sn;
        float *dst = L->ring[chunk & (kChunks - 1)][wave];
#pragma unroll
        for (int d = 0; d < D; ++d)
            __builtin_amdgcn_global_load_lds((knn_glb_void_t *)(src + d * sd), (knn_lds_void_t *)(dst + d * kRow), 4, 0, 0);
    }
    // (raw s_barrier: __syncthreads() would drain vmcnt(0), i.e. wait for the chunks just requested)
    __device__ __forceinline__ void start() const {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // nobody still reads the ring of the previous pass
#pragma unroll
        for (int t = 0; t < kChunksAhead; ++t) request(t);
    }
    // chunk t is complete in LDS for the whole block; chunk t + kChunksAhead is on its way
    __device__ __forceinline__ void acquire(int chunk) const {
        request(chunk + kChunksAhead);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kChunksAhead * D) : "memory");
    }
    __device__ __forceinline__ void finish() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    // distances of the 64 points of step `st` of `chunk` to the 16 queries (layout above)
    __device__ __forceinline__ knn_f32x16 step(int chunk, int st) const {
        const float *slot = L->ring[chunk & (kChunks - 1)][st];
        const int base = (chunk * kChunkSteps + st) * RPE_WAVE;
        const float *row = slot + lane;
        float p[3];
        p[0] = row[0];
        p[1] = D > 1 ? row[kRow] : 0.f;
        p[2] = D > 2 ? row[2 * kRow] : 0.f;
        const float pp = base + lane < M ? rpe_sqnorm<D>(p) : INFINITY;  // past the end: every distance +inf
        const float *ap = aoff >= 0 ? slot + aoff : L->ones + (lane & 15);
        const knn_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const knn_f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[0], qb, zero, 0, 0, 0);
        const knn_f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[16], qb, zero, 0, 0, 0);
        const knn_f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[32], qb, zero, 0, 0, 0);
        const knn_f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[48], qb, zero, 0, 0, 0);
        const knn_f32x8 d01 = __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7);
        const knn_f32x8 d23 = __builtin_shufflevector(d2, d3, 0, 1, 2, 3, 4, 5, 6, 7);
        const knn_f32x16 acc = __builtin_shufflevector(d01, d23, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        return __builtin_amdgcn_mfma_f32_16x16x1f32(pp, 1.0f, acc, 0, 0, 0);
    }
};

// compare-exchange: afterwards x <= y if asc, x >= y otherwise (equal values: either way)
__device__ __forceinline__ void knn_cex(float &x, float &y, bool asc) {
    const bool swap = (y < x) == asc;
    const float nx = swap ? y : x, ny = swap ? x : y;
    x = nx;
    y = ny;
}

// bitonic sort, ascending, of the 64 values {v[i] of lanes c, 16 + c, 32 + c, 48 + c}: element e = 16 g + i, for all 16 c at once
__device__ __forceinline__ void bitonic64_cols(knn_f32x16 &v, int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int s = 2; s <= 64; s <<= 1) {
#pragma unroll
        for (int j = s >> 1; j > 0; j >>= 1) {
            if (j < 16) {  // partner register i ^ j of the same lane
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (i & j) continue;
                    const bool asc = s < 16 ? (i & s) == 0 : s == 16 ? (g & 1) == 0 : s == 32 ? (g & 2) == 0 : true;  // bit s of e = 16 g + i
                    float x = v[i], y = v[i | j];
                    knn_cex(x, y, asc);
                    v[i] = x;
                    v[i | j] = y;
                }
            } else {  // partner lane l ^ j (j = 16: g ^ 1, j = 32: g ^ 2)
                const bool lower = (lane & j) == 0;
                const bool asc = s == 32 ? (g & 2) == 0 : true;
                const bool keep_min = lower == asc;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float x = v[i];
                    const float y = __shfl_xor(x, j);
                    v[i] = ((y < x) == keep_min) ? y : x;
                }
            }
        }
    }
}

// QUEUE_OUT: the workgroup's tied rows are not redone here but handed to knn_tie_replay_kernel (launched right behind this
// kernel) through the caller's workspace: count[block] and rows[block][.] are written by EVERY workgroup, so the workspace
// needs no initialisation.
struct TieOut {
    int *count;  // [B * ceil(Q / 64)]
    int *rows;   // [B * ceil(Q / 64)][64]: query indices
};
struct TieOuts {
    TieOut job[RPE_KNN_MAX_JOBS];
};

template <int D, bool QUEUE_OUT>
__global__ __launch_bounds__(kWavesPerBlock * RPE_WAVE) void knn_mfma_kernel(KnnJobs jobs, int k, int exact_ties, TieOuts touts) {
    __shared__ MfmaBlockLds lds;
    const rpe_knn_job &J = jobs.job[blockIdx.z];
    const float *__restrict__ inp = J.input;
    const float *__restrict__ qry = J.query;
    const int64_t in_sb = J.in_sb, in_sn = J.in_sn, in_sd = J.in_sd, q_sb = J.q_sb, q_sn = J.q_sn, q_sd = J.q_sd;
    const int M = J.M, Q = J.Q;
    int64_t *__restrict__ idx = J.idx;
    float *__restrict__ dist = J.dist;
    const int lane = rpe_lane(), g = lane >> 4, c = lane & 15;
    const int wave = rpe_uniform((int)(threadIdx.x >> 6));
    const int b = blockIdx.y;
    if ((int)blockIdx.x * kWavesPerBlock * kMq >= Q) return;  // whole block beyond this job's queries (block-uniform)
    const int qbase = (blockIdx.x * kWavesPerBlock + wave) * kMq;  // a wave beyond Q still loads its share of the cloud
    inp += (int64_t)b * in_sb;
    qry += (int64_t)b * q_sb;
    if (threadIdx.x < kRow) lds.r.ones[threadIdx.x] = 1.0f;
    __shared__ TieQueue tq;
    if (threadIdx.x == 0) tq.n = 0;  // (the sweeps' barriers come before the first push)

    MfmaSweep<D> sw;
    sw.inp = inp, sw.sn = in_sn, sw.sd = in_sd, sw.M = M, sw.lane = lane, sw.wave = wave, sw.L = &lds.r;
    float qv[3];  // lane c of every group: query qbase + c
    load_point<D>(qry, q_sn, q_sd, min(qbase + c, Q - 1), qv);
    sw.qb = g == 0 ? -2.0f * qv[0] : g == 1 ? -2.0f * qv[1] : g == 2 ? -2.0f * qv[2] : rpe_sqnorm<D>(qv);  // (a missing dimension is 1 * 0)
    sw.aoff = g < D ? g * kRow + c : -1;
    const int kk = (exact_ties && k < M && k < RPE_WAVE) ? k + 1 : k;
    const int n_chunks = (M + kChunkSteps * RPE_WAVE - 1) / (kChunkSteps * RPE_WAVE);

    // ---- pass A: 64 minima per query over disjoint subsets, their kk-th smallest
    knn_f32x16 lm;
#pragma unroll
    for (int i = 0; i < 16; ++i) lm[i] = INFINITY;
    sw.start();
    for (int t = 0; t < n_chunks; ++t) {
        sw.acquire(t);
#pragma unroll
        for (int st = 0; st < kChunkSteps; ++st) {
            const knn_f32x16 acc = sw.step(t, st);  // (steps past the end give +inf)
#pragma unroll
            for (int i = 0; i < 16; ++i) lm[i] = knn_min(lm[i], acc[i]);
        }
    }
    sw.finish();
    bitonic64_cols(lm, lane);
    float tau;
    {
        const int ei = (kk - 1) & 15;
        float sel = lm[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) sel = ei == i ? lm[i] : sel;
        const float bound = __shfl(sel, (((kk - 1) >> 4) << 4) | c);
        // strict comparison below: step to the next float up so that d == bound still passes
        const int bits = __float_as_int(bound);
        tau = !(bound < INFINITY) ? INFINITY : bound == 0.f ? __int_as_float(1) : bound > 0.f ? __int_as_float(bits + 1) : __int_as_float(bits - 1);
    }

    // ---- pass B: every lane collects the points below its query's bound (index order inside a lane)
    // The entry is ALWAYS written to the slot behind the lane's last one and the list only grows on a hit: no divergent
    // code, no cross-lane traffic; a 16-point tile without a hit in any lane is skipped as a whole (one v_min3 + v_min).
    const unsigned list_base = (unsigned)(uintptr_t)&lds.list[wave][0][lane];  // LDS byte address of this lane's slot 0
    int cnt = 0;
    auto collect = [&](const knn_f32x16 &acc, int base) {
        float m4[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) m4[bb] = knn_min(knn_min3(acc[4 * bb], acc[4 * bb + 1], acc[4 * bb + 2]), acc[4 * bb + 3]);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            if (__ballot(m4[bb] < tau) == 0ull) continue;  // wave-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc[4 * bb + r];
                const unsigned addr = list_base + (unsigned)min(cnt, kLaneList - 1) * (unsigned)(RPE_WAVE * 8);
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(addr), "v"(d), "v"(base + 16 * bb + r) : "memory");
                cnt += d < tau ? 1 : 0;
            }
        }
    };
    sw.start();
    for (int t = 0; t < n_chunks; ++t) {
        sw.acquire(t);
        // the matrix instructions of the next step are issued before the current step's results are looked at
        knn_f32x16 cur = sw.step(t, 0);
#pragma unroll
        for (int st = 0; st < kChunkSteps; ++st) {
            knn_f32x16 nxt = cur;
            if (st + 1 < kChunkSteps) nxt = sw.step(t, st + 1);
            collect(cur, (t * kChunkSteps + st) * RPE_WAVE + 4 * g);
            cur = nxt;
        }
    }
    sw.finish();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the asm stores above are invisible to the compiler's counters
    lds.count[wave][lane] = cnt;

    // ---- rank the collected entries, all 16 queries at once.  key = (orderable distance bits, index): the order the
    // insertion sweep produces.  (1) every lane moves its entries to a dense per-query array; (2) lane (g, c) takes entries
    // g, g + 4, ... of query c and counts the entries of that query with a smaller key; (3) rank r goes to out[c][r].  dense
    // and out both live in the list memory (a lane holds what it still needs in registers; one wave, LDS in order).
    const int have = min(cnt, kLaneList - 1);
    int have_g[4];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) have_g[gg] = min(lds.count[wave][16 * gg + c], kLaneList - 1);
    const unsigned long long full_lanes = __ballot(cnt >= kLaneList);
    const int n_query = have_g[0] + have_g[1] + have_g[2] + have_g[3];  // (<= 60)
    int dense_at = 0;  // entries of this query in the lanes before this one
#pragma unroll
    for (int gg = 0; gg < 3; ++gg) dense_at += gg < g ? have_g[gg] : 0;
    int max_n = n_query;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) max_n = max(max_n, __shfl_xor(max_n, off));
    max_n = rpe_uniform(max_n);
    unsigned long long *const dense = &lds.list[wave][0][0] + c * RPE_WAVE;  // this lane's query: [64]
    {
        unsigned long long own[kLaneList - 1];
#pragma unroll
        for (int sl = 0; sl < kLaneList - 1; ++sl) own[sl] = lds.list[wave][sl][lane];
#pragma unroll
        for (int sl = 0; sl < kLaneList - 1; ++sl)
            if (sl < have) dense[dense_at + sl] = own[sl];
    }
    auto key_of = [](unsigned long long e) {  // (index << 32 | distance bits) -> (orderable distance << 32 | index)
        const unsigned db = (unsigned)e, ix = (unsigned)(e >> 32);
        const unsigned ord = db ^ (((int)db >> 31) | 0x80000000u);  // distances are never -0 (|p|^2, |q|^2 >= +0)
        return ((unsigned long long)ord << 32) | ix;
    };
    auto rank_all = [&](auto ns_tag) {
        constexpr int NS = decltype(ns_tag)::value;  // entries per lane: ceil(max_n / 4)
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
            if (g + 4 * j < n_query) dense[rank[j]] = mine_e[j];  // ranks are a permutation of 0 .. n_query - 1
    };
    if (max_n <= 16) rank_all(std::integral_constant<int, 4>{});
    else if (max_n <= 32) rank_all(std::integral_constant<int, 8>{});
    else rank_all(std::integral_constant<int, 16>{});

    for (int q = 0; q < kMq; ++q) {
        const int qi = qbase + q;
        if (qi >= Q) break;  // wave-uniform
        float qc[3], qm2[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) qc[d] = rpe_readlane(qv[d], q);
        const float qq = rpe_sqnorm<D>(qc);
#pragma unroll
        for (int d = 0; d < 3; ++d) qm2[d] = -2.0f * qc[d];
        float Ld;
        int Li;
        const int n = rpe_readlane(n_query, q);
        if (((full_lanes >> q) & 0x0001000100010001ull) != 0ull || n > RPE_WAVE) {  // a lane list of this query filled up
            if (QUEUE_OUT && exact_ties == RPE_KNN_TIES_TORCH) {
                // torch.topk's order is what the replay kernel computes for ANY row (it is the reference's algorithm on the
                // row's distances): the row goes there instead of through the serial sweep, whose wave ran on alone after
                // every other wave of the launch had finished (+12 us on 8 x (8192 -> 4096), k = 16)
                if (lane == 0) tq.query[atomicAdd(&tq.n, 1)] = qi;
                continue;
            }
            serial_select<D>(inp, in_sn, in_sd, M, qm2, qq, kk, lane, Ld, Li);
        } else {
            const unsigned long long e = lds.list[wave][0][q * RPE_WAVE + lane];
            Ld = lane < n ? __int_as_float((int)(unsigned)e) : INFINITY;
            Li = lane < n ? (int)(e >> 32) : 0;
        }
        if (has_ties(Ld, k, kk, exact_ties, lane)) {
            if (lane == 0) tq.query[atomicAdd(&tq.n, 1)] = qi;
        } else if (lane < k) {
            const int64_t o = ((int64_t)b * Q + qi) * k + lane;
            idx[o] = (int64_t)Li;
            if (dist) dist[o] = Ld;
        }
    }
    if constexpr (QUEUE_OUT) {
        __syncthreads();
        const TieOut &T = touts.job[blockIdx.z];
        const int blk = b * ((Q + kWavesPerBlock * kMq - 1) / (kWavesPerBlock * kMq)) + blockIdx.x;
        const int n = tq.n;
        if (threadIdx.x == 0) T.count[blk] = n;
        if ((int)threadIdx.x < n) T.rows[blk * (kWavesPerBlock * kMq) + threadIdx.x] = tq.query[threadIdx.x];
    } else {
        if (exact_ties) drain_ties<D>(&tq, J, b, k, lane, wave);
    }
}

// ---- the matrix kernel's tied rows, redone by a second launch ------------------------------------------------------------
// In knn_mfma_kernel a tied row is replayed by one wave after the workgroup's sweeps, and the launch ends with the last such
// wave (~45 us behind a 124 us sweep on 8 x (8192 -> 4096), k = 16).  Here the tied rows of the WHOLE launch (~1.4 % of the
// rows of a real cloud) are shared out over a grid of their own, one row per workgroup at a time: its four waves first put the
// row's M distances into LDS together (the scan that took the lone wave 32 load-wait-compare rounds), then one wave replays
// libstdc++'s __heap_select / __sort_heap on them exactly as heap_replay_static does -- same distances, same order, same sifts.
// Every workgroup scans the per-workgroup counts the sweep kernel left (at most kReplayMaxBlocks of them) and takes the rows
// t = blockIdx.x, blockIdx.x + gridDim.x, ... of the concatenated queues: no atomics, no initialised memory, a fixed order.
constexpr int kReplayMaxM = 16384;
constexpr int kReplayMaxBlocks = 4096;
constexpr int kReplayGrid = 1024;
constexpr int kReplayThreads = 256;

template <int K>
__device__ __forceinline__ void heap_replay_lds(const float *ld, int M, int lane, float &Ld, int &Li) {
    unsigned kv = heap_key(INFINITY);
    int iv = 0;
    float top = INFINITY;
    constexpr int kTiles = 4;
    for (int base0 = 0; base0 < M; base0 += kTiles * RPE_WAVE) {
        float d[kTiles];
        unsigned long long m[kTiles];
#pragma unroll
        for (int u = 0; u < kTiles; ++u) {
            const int i = base0 + u * RPE_WAVE + lane;
            d[u] = i < M ? ld[i] : INFINITY;
        }
        if (base0 == 0) {
            kv = heap_key(d[0]);
            iv = lane;
            if constexpr (K >= 2) heap_make<K, (K - 2) / 2>(kv, iv);
            top = heap_dist((unsigned)__builtin_amdgcn_readlane((int)kv, 0));
        }
        unsigned long long any = 0ull;
#pragma unroll
        for (int u = 0; u < kTiles; ++u) {
            m[u] = __ballot((base0 > 0 || u > 0 || lane >= K) && d[u] < top);
            any |= m[u];
        }
        if (!any) continue;
#pragma unroll
        for (int u = 0; u < kTiles; ++u) {
            unsigned long long mu = m[u] & __ballot(d[u] < top);
            while (mu) {
                const int l = __builtin_ctzll(mu);
                heap_sift<0, K>(kv, iv, heap_key(rpe_readlane(d[u], l)), base0 + u * RPE_WAVE + l);
                top = heap_dist((unsigned)__builtin_amdgcn_readlane((int)kv, 0));
                mu &= (~1ull << l) & __ballot(d[u] < top);
            }
        }
    }
    if constexpr (K >= 2) heap_sort<K - 1>(kv, iv);
    Ld = heap_dist(kv);
    Li = iv;
}

// the same for any k (LaneHeap: dynamic lanes)
__device__ __forceinline__ void heap_replay_lds_any(const float *ld, int M, int k, int lane, float &Ld, int &Li) {
    LaneHeap h;
    h.v = INFINITY;
    h.i = 0;
    float top = INFINITY;
    for (int base = 0; base < M; base += RPE_WAVE) {
        const float d = base + lane < M ? ld[base + lane] : INFINITY;
        if (base == 0) {
            h.v = d;
            h.i = lane;
            for (int parent = (k - 2) / 2;; --parent) {
                h.adjust(parent, k, h.val(parent), rpe_readlane(h.i, parent), lane);
                if (parent == 0) break;
            }
            top = h.val(0);
        }
        unsigned long long mu = __ballot((base > 0 || lane >= k) && d < top);
        while (mu) {
            const int l = __builtin_ctzll(mu);
            h.adjust(0, k, rpe_readlane(d, l), base + l, lane);
            top = h.val(0);
            mu &= (~1ull << l) & __ballot(d < top);
        }
    }
    for (int last = k - 1; last >= 1; --last) {
        const float lv = h.val(last);
        const int li = rpe_readlane(h.i, last);
        h.move(last, 0, lane);
        h.adjust(0, last, lv, li, lane);
    }
    Ld = h.v;
    Li = h.i;
}

template <int D>
__global__ __launch_bounds__(kReplayThreads) void knn_tie_replay_kernel(KnnJobs jobs, TieOuts touts, int B, int k) {
    extern __shared__ float replay_lds[];  // [nblk + 1] ints: exclusive prefix of the counts; then M distances
    __shared__ int wave_sum[kReplayThreads / RPE_WAVE];
    const rpe_knn_job &J = jobs.job[blockIdx.y];
    const TieOut &T = touts.job[blockIdx.y];
    const int M = J.M, Q = J.Q;
    const int per_block = kWavesPerBlock * kMq;
    const int nbx = (Q + per_block - 1) / per_block, nblk = nbx * B;
    int *prefix = reinterpret_cast<int *>(replay_lds);
    float *ld = replay_lds + ((nblk + 1 + 3) & ~3);
    const int tid = threadIdx.x, lane = rpe_lane(), wave = tid >> 6;

    // exclusive scan of count[0 .. nblk): every thread owns `per` consecutive entries
    const int per = (nblk + kReplayThreads - 1) / kReplayThreads;  // <= 16
    int sum = 0;
    for (int j = 0; j < per; ++j) {
        const int c = tid * per + j;
        sum += c < nblk ? T.count[c] : 0;
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < RPE_WAVE; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == RPE_WAVE - 1) wave_sum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wave_sum[w];
    int total = 0;
    for (int w = 0; w < kReplayThreads / RPE_WAVE; ++w) total += wave_sum[w];
    for (int j = 0; j < per; ++j) {
        const int c = tid * per + j;
        if (c < nblk) {
            prefix[c] = run;
            run += T.count[c];
        }
    }
    if (tid == 0) prefix[nblk] = total;
    __syncthreads();

    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        int lo = 0, hi = nblk;  // the source workgroup: the last one whose prefix is <= t
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (prefix[mid] <= t) lo = mid;
            else hi = mid;
        }
        const int src = rpe_uniform(lo);
        const int qi = rpe_uniform(T.rows[src * per_block + (t - prefix[src])]);
        const int b = src / nbx;
        const float *inp = J.input + (int64_t)b * J.in_sb;
        float q[3], qm2[3];
        load_point<D>(J.query + (int64_t)b * J.q_sb, J.q_sn, J.q_sd, qi, q);
#pragma unroll
        for (int d = 0; d < 3; ++d) q[d] = rpe_uniform(q[d]);
        const float qq = rpe_sqnorm<D>(q);
#pragma unroll
        for (int d = 0; d < 3; ++d) qm2[d] = -2.0f * q[d];
        constexpr int kAhead = 16;  // points per thread requested together: a row of 8192 is two memory round trips for the workgroup
        for (int i0 = tid; i0 < M; i0 += kAhead * kReplayThreads) {
            float p[kAhead][3];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) load_point<D>(inp, J.in_sn, J.in_sd, min(i0 + u * kReplayThreads, M - 1), p[u]);
#pragma unroll
            for (int u = 0; u < kAhead; ++u)
                if (i0 + u * kReplayThreads < M) ld[i0 + u * kReplayThreads] = rpe_pair_dist<D>(qm2, qq, p[u], rpe_sqnorm<D>(p[u]));
        }
        __syncthreads();
        if (wave == 0) {
            float Ld = 0.f;
            int Li = 0;
#if defined(RPE_REPLAY_PROBE)  // timing probe, never in the library: everything but the heap replay (7 of the kernel's 45 us)
            Ld = ld[lane];
#else
            // (the heap in SCALAR registers -- a decision tree of s_cmp / s_mov over compile-time positions -- was tried here, where
            // the wave holds nothing else: 522 SGPR spills at the tree's joins, 80 us against 45)
            if (k == 16) heap_replay_lds<16>(ld, M, lane, Ld, Li);
            else if (k == 3) heap_replay_lds<3>(ld, M, lane, Ld, Li);
            else heap_replay_lds_any(ld, M, k, lane, Ld, Li);
#endif
            if (lane < k) {
                const int64_t o = ((int64_t)b * Q + qi) * k + lane;
                J.idx[o] = (int64_t)Li;
                if (J.dist) J.dist[o] = Ld;
            }
        }
        __syncthreads();  // (the distances are overwritten by the next row)
    }
}


// ---- k == 1 on large clouds, many queries: the same matrix sweep, a running (minimum, index) per lane ------------------------
// A lane sees its query's points in index order (register 4 b + r = point 16 b + 4 g + r of the step), so a strict '<'
// keeps the first index inside the lane; the four lanes of a query then merge by (distance, index).  5 matrix + ~50 vector
// instructions per 64 points x 16 queries against ~130 vector instructions in knn_nearest_kernel.
template <int D>
__global__ __launch_bounds__(kWavesPerBlock * RPE_WAVE) void knn_mfma_nearest_kernel(KnnJobs jobs) {
    __shared__ MfmaRing lds;
    const rpe_knn_job &J = jobs.job[blockIdx.z];
    const int M = J.M, Q = J.Q;
    const int lane = rpe_lane(), g = lane >> 4, c = lane & 15;
    const int wave = rpe_uniform((int)(threadIdx.x >> 6));
    const int b = blockIdx.y;
    if ((int)blockIdx.x * kWavesPerBlock * kMq >= Q) return;  // whole block beyond this job's queries (block-uniform)
    const int qbase = (blockIdx.x * kWavesPerBlock + wave) * kMq;  // a wave beyond Q still loads its share of the cloud
    const float *inp = J.input + (int64_t)b * J.in_sb, *qry = J.query + (int64_t)b * J.q_sb;
    if (threadIdx.x < kRow) lds.ones[threadIdx.x] = 1.0f;

    MfmaSweep<D> sw;
    sw.inp = inp, sw.sn = J.in_sn, sw.sd = J.in_sd, sw.M = M, sw.lane = lane, sw.wave = wave, sw.L = &lds;
    float qv[3];
    load_point<D>(qry, J.q_sn, J.q_sd, min(qbase + c, Q - 1), qv);
    sw.qb = g == 0 ? -2.0f * qv[0] : g == 1 ? -2.0f * qv[1] : g == 2 ? -2.0f * qv[2] : rpe_sqnorm<D>(qv);
    sw.aoff = g < D ? g * kRow + c : -1;
    const int n_chunks = (M + kChunkSteps * RPE_WAVE - 1) / (kChunkSteps * RPE_WAVE);

    float bd = INFINITY;
    int bi = 0x7fffffff;
    sw.start();
    for (int t = 0; t < n_chunks; ++t) {
        sw.acquire(t);
#pragma unroll
        for (int st = 0; st < kChunkSteps; ++st) {
            const int base = (t * kChunkSteps + st) * RPE_WAVE + 4 * g;
            const knn_f32x16 acc = sw.step(t, st);  // (points past the end: +inf, never taken)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool take = acc[i] < bd;  // strict: first index wins inside a lane
                bd = take ? acc[i] : bd;
                bi = take ? base + 16 * (i >> 2) + (i & 3) : bi;
            }
        }
    }
    sw.finish();
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const float od = __shfl_xor(bd, off);
        const int oi = __shfl_xor(bi, off);
        const bool take = (od < bd) || (od == bd && oi < bi);
        bd = take ? od : bd;
        bi = take ? oi : bi;
    }
    const int qi = qbase + c;
    if (g == 0 && qi < Q) {
        const int64_t o = (int64_t)b * Q + qi;
        J.idx[o] = bi == 0x7fffffff ? 0 : (int64_t)bi;
        if (J.dist) J.dist[o] = bd;
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

#ifndef RPE_KNN_WAVE_TARGET
#define RPE_KNN_WAVE_TARGET 2048
#endif
// queries per wave of the insertion / lane-minimum kernels.  Two queries a wave is NOT offered to the selection kernel (k >= 2):
// measured (tools/knn_gate_table.py with RPE_KNN_WAVE_TARGET = 1024 / 2048 / 4096 builds, profiles/r05_knn_gate_table.txt) the
// QW = 2 instantiation is the slow one at every size it was picked for -- 4 x (1024 -> 1024), k = 16: 55.9 us against 36.6 with one
// query a wave and 45.5 with four; k = 3: 38.5 against 14.8 -- so a search too small for four queries a wave takes one.
int pick_qw(int B, int Q, int k) {
    const long target = RPE_KNN_WAVE_TARGET;  // waves wanted in flight: 256 CUs x 4 SIMDs x 2 (1024 ... 4096 measure the same; 8192 and more are slower)
    for (int qw = 8; qw > 1; qw >>= 1) {
        if (qw == 2 && k >= 2) continue;
        if ((long)B * ((Q + qw - 1) / qw) >= target) return qw;
    }
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
        // ... and two 16-bit position arrays for the wave-wide partition, while they fit
        const size_t narrow = (size_t)kWavesPerBlock * 2 * row * sizeof(float);
        size_t lds = narrow + (size_t)kWavesPerBlock * 2 * (row + 2) * sizeof(unsigned short);
        const int wide = lds <= 128 * 1024;
        if (!wide) lds = narrow;
        if (lds > 160 * 1024) return RPE_EUNSUPPORTED;
        if (lds > 64 * 1024) {  // beyond the default dynamic-LDS limit (k > 32 with a few thousand points)
            hipError_t e = hipFuncSetAttribute((const void *)knn_select_kernel<D, QW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((knn_select_kernel<D, QW, true>), grid, block, lds, st, jobs, k, ties, row, wide);
    } else {
        hipLaunchKernelGGL((knn_select_kernel<D, QW, false>), grid, block, 0, st, jobs, k, ties, 0, 0);
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

// the jobs of one launch group (all through the insertion / lane-minimum kernels, or all through the matrix kernel).
// ``ws``: per job, workspace for the matrix kernel's tied rows (all non-null: they are replayed by a second launch) or null.
int launch_group(const rpe_knn_job *const *jobs, char *const *ws, int njobs, bool matrix, int B, int D, int k, int tie_mode, hipStream_t st) {
    KnnJobs packed;
    int max_q = 0, min_m = 0x7fffffff, max_m = 0;
    long total_q = 0;
    bool replay = matrix && k >= 2 && tie_mode != RPE_KNN_TIES_INDEX && k < RPE_WAVE;
    TieOuts touts{};
    for (int i = 0; i < njobs; ++i) {
        const rpe_knn_job &j = *jobs[i];
        packed.job[i] = j;
        max_q = j.Q > max_q ? j.Q : max_q;
        min_m = j.M < min_m ? j.M : min_m;
        max_m = j.M > max_m ? j.M : max_m;
        total_q += j.Q;
        const int64_t nblk = (int64_t)B * ((j.Q + kWavesPerBlock * kMq - 1) / (kWavesPerBlock * kMq));
        replay = replay && ws && ws[i] && nblk <= kReplayMaxBlocks && j.M <= kReplayMaxM;
        if (replay) touts.job[i] = TieOut{reinterpret_cast<int *>(ws[i]), reinterpret_cast<int *>(ws[i]) + ((nblk + 3) & ~3ll)};
    }
    if (max_q == 0) return 0;
    if (matrix) {
        const int per_block = kWavesPerBlock * kMq;
        dim3 grid((max_q + per_block - 1) / per_block, B, njobs), block(kWavesPerBlock * RPE_WAVE);
        if (k == 1) {  // (M >= 64: topk's partial_sort with a one-element heap keeps the first minimum, as this kernel does)
            if (D == 3) hipLaunchKernelGGL(knn_mfma_nearest_kernel<3>, grid, block, 0, st, packed);
            else if (D == 2) hipLaunchKernelGGL(knn_mfma_nearest_kernel<2>, grid, block, 0, st, packed);
            else hipLaunchKernelGGL(knn_mfma_nearest_kernel<1>, grid, block, 0, st, packed);
            return rpe_launch_status();
        }
        if (!replay) {
            if (D == 3) hipLaunchKernelGGL((knn_mfma_kernel<3, false>), grid, block, 0, st, packed, k, tie_mode, touts);
            else if (D == 2) hipLaunchKernelGGL((knn_mfma_kernel<2, false>), grid, block, 0, st, packed, k, tie_mode, touts);
            else hipLaunchKernelGGL((knn_mfma_kernel<1, false>), grid, block, 0, st, packed, k, tie_mode, touts);
            return rpe_launch_status();
        }
        if (D == 3) hipLaunchKernelGGL((knn_mfma_kernel<3, true>), grid, block, 0, st, packed, k, tie_mode, touts);
        else if (D == 2) hipLaunchKernelGGL((knn_mfma_kernel<2, true>), grid, block, 0, st, packed, k, tie_mode, touts);
        else hipLaunchKernelGGL((knn_mfma_kernel<1, true>), grid, block, 0, st, packed, k, tie_mode, touts);
        int rc = rpe_launch_status();
        if (rc) return rc;
        const size_t lds = (((size_t)kReplayMaxBlocks + 4) + (size_t)max_m) * sizeof(float);
        const void *kern = D == 3 ? (const void *)knn_tie_replay_kernel<3> : D == 2 ? (const void *)knn_tie_replay_kernel<2> : (const void *)knn_tie_replay_kernel<1>;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        dim3 rgrid(kReplayGrid, njobs), rblock(kReplayThreads);
        if (D == 3) hipLaunchKernelGGL(knn_tie_replay_kernel<3>, rgrid, rblock, lds, st, packed, touts, B, k);
        else if (D == 2) hipLaunchKernelGGL(knn_tie_replay_kernel<2>, rgrid, rblock, lds, st, packed, touts, B, k);
        else hipLaunchKernelGGL(knn_tie_replay_kernel<1>, rgrid, rblock, lds, st, packed, touts, B, k);
        return rpe_launch_status();
    }
    const int qw = pick_qw(B, (int)total_q, k);
    if (D == 3) return launch_knn_d<3>(qw, packed, njobs, max_q, min_m, max_m, B, k, tie_mode, st);
    if (D == 2) return launch_knn_d<2>(qw, packed, njobs, max_q, min_m, max_m, B, k, tie_mode, st);
    return launch_knn_d<1>(qw, packed, njobs, max_q, min_m, max_m, B, k, tie_mode, st);
}

// which kernels take a search (the same rules size the workspace)
bool takes_matrix(int B, int M, int Q, int k, int mode = 0) {
    if (mode & RPE_KNN_ALGO_INSERT) return false;
    const bool can = k < RPE_WAVE && (long)M >= 64L * k && M >= 4 * RPE_WAVE;  // topk's partial_sort regime; one ring chunk of points
    if (mode & RPE_KNN_ALGO_MATRIX) return can;
    return can && M >= kMatrixMinM && (long)B * Q >= kMatrixMinQueries;
}
// the binned nearest-point search (knn_binned.hip: k = 1, D = 2) against the sweeps -- two launches (8 us to bin the clouds),
// then waves that meet ~50 points instead of the whole cloud.  Kernel time, both frames of a batch of 4: 4096 points /
// 144 x 240 queries 8 + 22 us against 201; 2048 / 72 x 120: 8 + 13 against 34; 1024 / 36 x 60: 8 + 10 against 10.
bool takes_binned(int B, int M, int Q, int D, int k) { return D == 2 && k == 1 && M >= 2048 && (long)B * Q >= 16384; }

int64_t job_workspace_bytes(int B, int M, int Q, int D, int k, int mode) {
    if (B <= 0 || M <= 0 || Q <= 0) return 0;
    if (D == 2 && k == 1 && M >= 64 && ((mode & RPE_KNN_ALGO_BINNED) || (takes_binned(B, M, Q, D, k) && !(mode & RPE_KNN_ALGO_SWEEP))))
        return rpe_nearest2d_workspace_bytes(B, M);
    if (k >= 2 && takes_matrix(B, M, Q, k, mode) && (mode & 3) != RPE_KNN_TIES_INDEX && M <= kReplayMaxM) {
        const int64_t nblk = (int64_t)B * ((Q + kWavesPerBlock * kMq - 1) / (kWavesPerBlock * kMq));
        if (nblk <= kReplayMaxBlocks) return ((((nblk + 3) & ~3ll) + nblk * (kWavesPerBlock * kMq)) * 4 + 15) & ~15ll;
    }
    return 0;
}

}  // namespace

RPE_API int64_t rpe_knn_workspace_bytes(int B, int M, int Q, int D, int k, int mode) { return job_workspace_bytes(B, M, Q, D, k, mode); }

RPE_API int rpe_knn_multi(const rpe_knn_job *jobs, int njobs, int B, int D, int k, int mode, void *workspace, int64_t workspace_bytes,
                          rpe_stream_t stream) {
    if (!jobs || njobs < 1 || njobs > RPE_KNN_MAX_JOBS || B < 0 || D < 1 || D > 3 || k < 1) return RPE_EINVAL;
    const int tie_mode = mode & 3;
    if ((mode & ~(3 | RPE_KNN_ALGO_SWEEP | RPE_KNN_ALGO_BINNED | RPE_KNN_ALGO_MATRIX | RPE_KNN_ALGO_INSERT)) || tie_mode == 2 ||
        ((mode & RPE_KNN_ALGO_SWEEP) && (mode & RPE_KNN_ALGO_BINNED)) || ((mode & RPE_KNN_ALGO_MATRIX) && (mode & RPE_KNN_ALGO_INSERT)))
        return RPE_EINVAL;
    if (k > RPE_WAVE) return RPE_EUNSUPPORTED;
    if (B > 65535) return RPE_EUNSUPPORTED;
    if (workspace && ((reinterpret_cast<uintptr_t>(workspace) & 15) || workspace_bytes < 0)) return RPE_EINVAL;
    // a cloud of at least kMatrixMinM points in topk's partial_sort regime (64 k <= M), enough queries: the matrix kernels;
    // k = 1, D = 2 on a large cloud with workspace: the binned search; everything else: insertion / lane-minimum kernels
    const rpe_knn_job *big[RPE_KNN_MAX_JOBS], *rest[RPE_KNN_MAX_JOBS], *binned[RPE_KNN_MAX_JOBS];
    char *big_ws[RPE_KNN_MAX_JOBS], *binned_ws[RPE_KNN_MAX_JOBS];
    int nbig = 0, nrest = 0, nbinned = 0;
    char *ws = static_cast<char *>(workspace);
    int64_t left = workspace ? workspace_bytes : 0;
    bool all_big_ws = true;
    for (int i = 0; i < njobs; ++i) {
        const rpe_knn_job &j = jobs[i];
        if (!j.input || !j.query || !j.idx || j.M <= 0 || j.Q < 0 || k > j.M) return RPE_EINVAL;
        const int64_t need = job_workspace_bytes(B, j.M, j.Q, D, k, mode);  // jobs take their shares in order
        char *mine = (need > 0 && left >= need) ? ws : nullptr;
        if (mine) ws += need, left -= need;
        const bool want_binned = D == 2 && k == 1 && j.M >= 64 && B > 0 && j.Q > 0 &&
                                 ((mode & RPE_KNN_ALGO_BINNED) || (takes_binned(B, j.M, j.Q, D, k) && !(mode & RPE_KNN_ALGO_SWEEP)));
        if ((mode & RPE_KNN_ALGO_BINNED) && B > 0 && j.Q > 0 && !(want_binned && mine)) return (D == 2 && k == 1 && j.M >= 64) ? RPE_EINVAL : RPE_EUNSUPPORTED;
        if (want_binned && mine) {
            binned[nbinned] = &j, binned_ws[nbinned++] = mine;
        } else if (takes_matrix(B, j.M, j.Q, k, mode)) {
            big[nbig] = &j, big_ws[nbig++] = mine;
            all_big_ws = all_big_ws && mine;
        } else {
            rest[nrest++] = &j;
        }
    }
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < nbinned; ++i) {
        const rpe_knn_job &j = *binned[i];
        const int rc = rpe_nearest2d(j.input, j.in_sb, j.in_sn, j.in_sd, j.query, j.q_sb, j.q_sn, j.q_sd, B, j.M, j.Q, j.idx, j.dist, binned_ws[i], st);
        if (rc) return rc;
    }
    if (nbig) {
        const int rc = launch_group(big, all_big_ws ? big_ws : nullptr, nbig, true, B, D, k, tie_mode, st);
        if (rc) return rc;
    }
    return nrest ? launch_group(rest, nullptr, nrest, false, B, D, k, tie_mode, st) : 0;
}

RPE_API int rpe_knn(const float *input, int64_t in_sb, int64_t in_sn, int64_t in_sd, const float *query, int64_t q_sb,
                    int64_t q_sn, int64_t q_sd, int B, int M, int Q, int D, int k, int mode, int64_t *idx, float *dist,
                    void *workspace, int64_t workspace_bytes, rpe_stream_t stream) {
    if (!input || !query || !idx || B < 0 || M <= 0 || Q < 0 || D < 1 || D > 3) return RPE_EINVAL;
    if (k < 1 || k > M) return RPE_EINVAL;
    const rpe_knn_job job{input, in_sb, in_sn, in_sd, query, q_sb, q_sn, q_sd, M, Q, idx, dist};
    return rpe_knn_multi(&job, 1, B, D, k, mode, workspace, workspace_bytes, stream);
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
