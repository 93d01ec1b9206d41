// ABI bookkeeping for librpeflow_hip.so.
#include "common.h"

RPE_API int rpe_abi_version(void) { return RPE_ABI_VERSION; }

RPE_API const char *rpe_error_string(int code) {
    if (code == 0) return "success";
    if (code == RPE_EINVAL) return "rpeflow_hip: invalid argument";
    if (code == RPE_EUNSUPPORTED) return "rpeflow_hip: unsupported size for this build";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "rpeflow_hip: unknown error";
}

namespace {
// s_memtime (the free-running counter of the shader engine clock) and s_memrealtime (the constant-rate reference clock) read
// back to back by one wave: two such pairs bracket a stretch of a stream, d(shader) / d(wall) x the wall rate = the clock
// the engines actually ran at over that stretch
__global__ void clock_stamp_kernel(unsigned long long *slot) {
    slot[0] = clock64();
    slot[1] = wall_clock64();
}
}  // namespace

RPE_API int rpe_clock_stamp(unsigned long long *slot2, int *wall_khz, rpe_stream_t stream) {
    if (!slot2) return RPE_EINVAL;
    if (wall_khz) {
        int device = 0;
        hipError_t e = hipGetDevice(&device);
        if (e == hipSuccess) e = hipDeviceGetAttribute(wall_khz, hipDeviceAttributeWallClockRate, device);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot2);
    return rpe_launch_status();
}
