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
__global__ void stamp_kernel(unsigned long long *slot) { *slot = wall_clock64(); }
}  // namespace

RPE_API int rpe_debug_stamp(unsigned long long *slot, rpe_stream_t stream) {
    if (!slot) return RPE_EINVAL;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot);
    return rpe_launch_status();
}
