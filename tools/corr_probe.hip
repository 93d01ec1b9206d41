// Timing probes for corr_mfma_dma_kernel on BASELINE config 2 (1x256x544x960, N(0,1) inputs): what each part of the
// kernel costs on its own.  Build one binary per probe and run it on the GPU (120 warm-up launches: the clocks need
// ~40 ms of load to settle, then 20 timed):
//   for m in 0 1 2 8 10 12; do hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DRPE_CORR_PROBE=$m \
//       -I include -I rpeflow_amd/csrc -o /tmp/corr_probe$m tools/corr_probe.hip rpeflow_amd/csrc/abi.hip; done
// RPE_CORR_PROBE values are listed in correlation.hip.  Results are wrong by construction for every value but 0.
#include "../rpeflow_amd/csrc/correlation.hip"
#include <cstdio>
#include <vector>
#include <random>
int main(int argc, char **argv) {
    const int B = 1, C = 256, H = 544, W = 960;
    int algo = argc > 1 ? atoi(argv[1]) : 7;
    size_t n = (size_t)B * C * H * W, no = (size_t)B * 81 * H * W;
    float *a, *b, *o;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o, no * 4);
    std::vector<float> h(n);
    std::mt19937 gen(1); std::normal_distribution<float> nd(0.f, 1.f);
    const float scale = argc > 2 ? (float)atof(argv[2]) : 1.f;  // 0: all-zero inputs (no operand toggling: the power-unlimited time)
    for (size_t i = 0; i < n; ++i) h[i] = nd(gen) * scale;
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(b, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 120; ++i) rpe_correlation2d_forward(a, b, B, C, H, W, 4, 0.f, algo, o, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < 20; ++i) rpe_correlation2d_forward(a, b, B, C, H, W, 4, 0.f, algo, o, nullptr);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("probe %d algo %d: %.1f us\n", RPE_CORR_PROBE, algo, ms * 50.f);
    return 0;
}
