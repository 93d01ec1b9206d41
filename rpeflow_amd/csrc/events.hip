// Event voxelisation on the device (event_utils.py:109-128 eventsToVoxel -> :211-262 events_to_voxel_torch with
// temporal_bilinear=True, :264-303 the positive / negative split).  The reference does this
// in the dataloader on the CPU, once per bin with index_put_(accumulate=True); here one launch handles all bins.
//
// To reproduce the reference's fp32 sums bit for bit the events arrive STABLY SORTED BY PIXEL (the host wrapper sorts
// with torch): one thread owns one pixel and adds its events in their original order, exactly the order in which the
// CPU's index_put_ accumulates them, so no atomics and no run-to-run variation.  The time normalisation is the
// reference's two-step float64 arithmetic (eventsToXYTP :31-35, then :243-244); a weight max(0, 1 - |t - bin|) * p is
// rounded to fp32 before it is added, as `.float()` does at :207.
#include "common.h"

namespace {

// out [C][HW] zero-initialised, C = bins (signed polarity weights) or 2 * bins (positive grids, then negative grids).
// T = double: float64 event arrays (the arithmetic numpy / torch do on them).  T = float: the float32 [N,4] arrays
// load_events_h5 returns (event_utils.py:11-20) -- every step below is then a float32 operation, rounded where numpy and torch
// round it (the 1e-6 joins deltaT as a float32; 1.0 - |t - bin| and the product with the polarity are float32).
template <typename T>
__global__ __launch_bounds__(256) void events_to_voxel_kernel(const int *__restrict__ pixel, const T *__restrict__ t,
                                                              const int *__restrict__ pol, const int *__restrict__ run_start, int runs,
                                                              int n_events, T t_first, T t_last, int bins, int split, int64_t HW,
                                                              float *__restrict__ out) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= runs) return;
    const int begin = run_start[u], end = u + 1 < runs ? run_start[u + 1] : n_events;
    const int p = pixel[begin];
    const T span = (t_last - t_first) + (T)1e-6;               // eventsToXYTP: (t - t0) / (deltaT + 1e-6)
    const T te_last = (t_last - t_first) / span, te_first = (t_first - t_first) / span;
    const T dt = te_last - te_first;                           // events_to_voxel_torch: dt = ts[-1] - ts[0]
    for (int i = begin; i < end; ++i) {
        const T te = (t[i] - t_first) / span;
        const T tn = (te - te_first) / dt * (T)(bins - 1);     // t_norm, left to right as written at :244
        const int sign = pol[i];
        if (tn != tn) {  // all timestamps equal (dt = 0 -> 0/0) or a NaN timestamp: the reference's weight max(0, 1 - |t_norm - b|) is NaN
            const T nan_w = tn;  // in EVERY bin, and polarity * NaN (0 * NaN in the other polarity's grid) is NaN too
            for (int c = 0; c < (split ? 2 * bins : bins); ++c) out[(int64_t)c * HW + p] += (float)nan_w;
            continue;
        }
        const int b0 = (int)floor(tn);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int b = b0 + k;
            if (b < 0 || b >= bins) continue;
            const T d = tn - (T)b;
            const T w = (T)1 - (d < (T)0 ? -d : d);
            const T wz = w < (T)0 ? (T)0 : w;  // torch.max(zeros, w): a NaN weight (all timestamps equal: t_norm = 0/0) stays NaN
            if (split) {  // :296-297: positive grid takes p > 0, negative grid p <= 0, each with weight 1
                float *o = out + (int64_t)(sign > 0 ? b : bins + b) * HW + p;
                *o = *o + (float)((T)1 * wz);
            } else {
                float *o = out + (int64_t)b * HW + p;
                *o = *o + (float)((T)sign * wz);
            }
        }
    }
}

template <typename T>
int launch_events(const int *pixel_sorted, const T *t_sorted, const int *polarity_sorted, const int *run_start, int runs, int n_events,
                  T t_first, T t_last, int bins, int split_polarity, int64_t HW, float *out, rpe_stream_t stream) {
    if (!pixel_sorted || !t_sorted || !polarity_sorted || !run_start || !out || runs < 0 || n_events < 0 || bins < 1 || HW < 1)
        return RPE_EINVAL;
    if (runs == 0 || n_events == 0) return 0;
    hipLaunchKernelGGL(events_to_voxel_kernel<T>, dim3((runs + 255) / 256), dim3(256), 0, (hipStream_t)stream, pixel_sorted, t_sorted,
                       polarity_sorted, run_start, runs, n_events, t_first, t_last, bins, split_polarity, HW, out);
    return rpe_launch_status();
}

}  // namespace

RPE_API int rpe_events_to_voxel(const int *pixel_sorted, const void *t_sorted, int t_is_f32, const int *polarity_sorted,
                                const int *run_start, int runs, int n_events, double t_first, double t_last, int bins,
                                int split_polarity, int64_t HW, float *out, rpe_stream_t stream) {
    if (t_is_f32)
        return launch_events<float>(pixel_sorted, (const float *)t_sorted, polarity_sorted, run_start, runs, n_events, (float)t_first,
                                    (float)t_last, bins, split_polarity, HW, out, stream);
    return launch_events<double>(pixel_sorted, (const double *)t_sorted, polarity_sorted, run_start, runs, n_events, t_first, t_last, bins,
                                 split_polarity, HW, out, stream);
}
