// Diagnostics (not on the product path): what the chip can actually sustain.
//   nsg_debug_mfma_peak : a register-only v_mfma_f32_32x32x2_f32 loop on random data, every CU busy --
//                         the fp32 matrix rate at the clock the chip holds under that load, and that clock
//                         (s_memtime ticks are shader cycles, s_memrealtime ticks are 100 MHz).
#include "nsg_common.h"

namespace {

__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float seed, float *sink, unsigned long long *stamps)
{
    v16f acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed * (float)(threadIdx.x % 17 + 1) * 1e-3f, b = seed * (float)(threadIdx.x % 13 + 1) * -1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
        }
        a += 1e-6f;
        b -= 1e-6f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

}  // namespace

extern "C" {

// sink: blocks*256 floats, stamps: blocks*2 u64 (shader-cycle delta, 100 MHz-tick delta per block)
NSG_API int nsg_debug_mfma_peak(int32_t blocks, int32_t iters, float *sink, unsigned long long *stamps, void *stream)
{
    NSG_REQUIRE(blocks > 0 && iters > 0 && sink && stamps, NSG_E_INVALID, "nsg_debug_mfma_peak: bad argument");
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, 1.0f, sink, stamps);
    return nsg_check_launch("mfma_peak_kernel");
}

}  // extern "C"
