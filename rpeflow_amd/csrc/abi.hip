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
// s_memtime (the free-running counter of an XCD's engine clock) and s_memrealtime (the constant-rate reference clock) read back to
// back, ONCE PER XCD: sixteen one-wave workgroups are dealt round robin over the eight XCDs, each stores its pair in the slot of
// the XCD it runs on (HW_REG_XCC_ID).  The cycle counters of different XCDs are not one clock -- an XCD that idles while the
// others work (an 8-workgroup launch, a thin tail) stops counting -- so a stretch is measured per XCD, from that XCD's own two
// stamps: d(shader) / d(wall) x the wall rate = the clock that XCD ran at.
__global__ void clock_stamp_kernel(unsigned long long *slot) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    if (threadIdx.x == 0) {
        slot[2 * xcc] = clock64();
        slot[2 * xcc + 1] = wall_clock64();
    }
}
}  // namespace

RPE_API int rpe_clock_stamp(unsigned long long *slot16, int *wall_khz, rpe_stream_t stream) {
    if (!slot16) return RPE_EINVAL;
    if (wall_khz) {
        int device = 0;
        hipError_t e = hipGetDevice(&device);
        if (e == hipSuccess) e = hipDeviceGetAttribute(wall_khz, hipDeviceAttributeWallClockRate, device);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(16), dim3(64), 0, (hipStream_t)stream, slot16);
    return rpe_launch_status();
}
