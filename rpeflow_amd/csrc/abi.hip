// ABI bookkeeping for librpeflow_hip.so.
#include "common.h"

// lower 16 bits: RPE_ABI_VERSION; upper 16: non-zero only in a diagnostic build that computes wrong results on purpose
RPE_API int rpe_abi_version(void) { return RPE_ABI_VERSION | (rpe_diagnostic_flavour() << 16); }

RPE_API const char *rpe_error_string(int code) {
    if (code == 0) return "success";
    if (code == RPE_EINVAL) return "rpeflow_hip: invalid argument";
    if (code == RPE_EUNSUPPORTED) return "rpeflow_hip: unsupported size for this build";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "rpeflow_hip: unknown error";
}

namespace {
// one thread stores the engine cycle counter and the constant-rate counter: a marker for timelines (the constant-rate entry)
__global__ void clock_stamp_kernel(unsigned long long *slot) {
    slot[0] = clock64();
    slot[1] = wall_clock64();
}

// The engine clock over a stretch of a stream, from two of these stamps: a few thousand one-wave workgroups cover the chip,
// each stores s_memtime (engine cycles) and s_memrealtime (constant rate) in the slot of the compute unit it runs on
// (HW_REG_XCC_ID, and the shader-engine / array / CU fields of HW_REG_HW_ID: 8 x 256 slots).  A slot is only ever compared
// with itself: the cycle counters of different parts of the chip are not one clock -- two stamps taken wherever a one-workgroup
// kernel landed read 1455 "MHz" after 8-workgroup launches had let counters drift apart, one slot per XCD gave negative
// differences -- and under the package power limit the XCDs do not run at one clock either.  (Probes that run BESIDE the
// measured launches were tried: they take compute units and a hardware queue from them.)
__global__ __launch_bounds__(64) void clock_stamp_all_kernel(unsigned long long *slots) {
    if (threadIdx.x != 0) return;
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned key = ((xcc & 7u) << 8) | ((hw >> 8) & 0xffu);  // HW_ID[15:8]: CU_ID, SH_ID, SE_ID
    // many of the 8192 workgroups land on one compute unit: the pair goes out as ONE 16-byte store, so a slot always holds
    // the two counters of one wave (two 8-byte stores of different waves could interleave)
    ulonglong2 pair;
    pair.x = clock64();
    pair.y = wall_clock64();
    *reinterpret_cast<ulonglong2 *>(slots + 2 * key) = pair;
}
}  // namespace

RPE_API int rpe_clock_stamp(unsigned long long *slot2, rpe_stream_t stream) {
    if (!slot2) return RPE_EINVAL;
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot2);
    return rpe_launch_status();
}

RPE_API int rpe_clock_stamp_all(unsigned long long *slots, int *wall_khz, rpe_stream_t stream) {
    if (!slots || (reinterpret_cast<uintptr_t>(slots) & 15)) return RPE_EINVAL;  // (16-byte stores)
    if (wall_khz) {
        int device = 0;
        hipError_t e = hipGetDevice(&device);
        if (e == hipSuccess) e = hipDeviceGetAttribute(wall_khz, hipDeviceAttributeWallClockRate, device);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(clock_stamp_all_kernel, dim3(8192), dim3(64), 0, (hipStream_t)stream, slots);
    return rpe_launch_status();
}
