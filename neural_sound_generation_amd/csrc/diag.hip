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

// The bf16 implicit-GEMM main loop with its memory side removed: per 128 x 128 x 64 chunk every thread writes 8 staged
// 16-byte pieces to LDS (register contents, no global loads), the workgroup barriers, and every wave reads its 16
// fragments and issues its 16 v_mfma_f32_32x32x16_bf16 -- exactly gather_gemm's LDS / barrier / MFMA traffic (pitch 36
// floats, double-buffered, two workgroups per CU).  Its rate is the ceiling the LDS-staged 128 x 128 design can reach.
// mode 0: stores + barrier + reads + MFMA; 1: no stores (reads + MFMA + barrier); 2: MFMA only.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void lds_fed_loop_kernel(int chunks, int mode, float *sink)
{
    constexpr int PITCH = 36;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][256 rows][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int rsub = tid >> 3, piece = (tid & 7) * 4;
    v4f stage[8];
    for (int j = 0; j < 8; ++j) stage[j] = v4f{(float)tid * 1e-3f, (float)j, 1.f, -1.f};
    v16f acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int i = tid; i < 2 * 256 * PITCH; i += 256) smem[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    for (int it = 0; it < chunks; ++it) {
        const int cur = it & 1;
        const float *a_base = smem + cur * 256 * PITCH + (wr * 64 + l31) * PITCH + 4 * h;
        const float *b_base = smem + cur * 256 * PITCH + (128 + wc * 64 + l31) * PITCH + 4 * h;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            v4f a[2], b[2];
            if (mode < 2) {
                for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const v4f *>(a_base + i * 32 * PITCH + kk * 8);
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const v4f *>(b_base + j * 32 * PITCH + kk * 8);
            } else {
                a[0] = a[1] = stage[kk]; b[0] = b[1] = stage[kk + 4];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        }
        if (mode == 0) {
            float *dst = smem + (cur ^ 1) * 256 * PITCH;
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<v4f *>(dst + (rsub + 32 * j) * PITCH + piece) = stage[j];
        }
        if (mode < 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    sink[blockIdx.x * 256 + tid] = s;
}

// The same loop with the staging done by LDS-DMA (buffer_load_dwordx4 ... lds: global -> LDS without the VGPR -> LDS transfer).
// A wave-instruction deposits 64 x 16 B contiguously, so rows are unpadded (128 B) and the fragment reads are XOR-swizzled
// (piece ^= (row >> 1) & 7, with row & 1 selecting the bank half) to stay conflict-free.  src: any 64 KiB (L2-resident).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void lds_dma_loop_kernel(int chunks, const float *src, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][256 rows][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 65536, 0x00020000);
    v16f acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int i = tid; i < 2 * 256 * 32; i += 256) smem[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int ra0 = wr * 64 + l31, rb0 = 128 + wc * 64 + l31;
    for (int it = 0; it < chunks; ++it) {
        const int cur = it & 1;
        const char *base = reinterpret_cast<const char *>(smem) + cur * 256 * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            v4f a[2], b[2];
            const int c = 2 * kk + h;
#pragma unroll
            for (int i = 0; i < 2; ++i) { const int r = ra0 + 32 * i; a[i] = *reinterpret_cast<const v4f *>(base + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)); }
#pragma unroll
            for (int j = 0; j < 2; ++j) { const int r = rb0 + 32 * j; b[j] = *reinterpret_cast<const v4f *>(base + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        }
        // 8 pieces per thread: wave w, piece j deposits rows 32 j + 8 w .. + 7 of the other buffer
        char *dstb = reinterpret_cast<char *>(smem) + (cur ^ 1) * 256 * 128;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dstb + (32 * j + 8 * wave) * 128), 16,
                                                     ((it * 8 + j) * 4096 + lane * 16) & 0xffff, 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the DMA writes have landed
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    sink[blockIdx.x * 256 + tid] = s;
}

// The same with 8 waves on a 256 x 128 tile (one workgroup per CU, still two waves per SIMD, each wave 64 x 64): the weight
// tile is staged once for twice the rows: 6 stores per thread per chunk instead of 8.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void lds_fed_loop8_kernel(int chunks, int mode, float *sink)
{
    constexpr int PITCH = 36;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][384 rows][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;   // wr 0..3, wc 0..1
    const int rsub = tid >> 3, piece = (tid & 7) * 4;                          // rsub 0..63
    v4f stage[6];
    for (int j = 0; j < 6; ++j) stage[j] = v4f{(float)tid * 1e-3f, (float)j, 1.f, -1.f};
    v16f acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int i = tid; i < 2 * 384 * PITCH; i += 512) smem[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    for (int it = 0; it < chunks; ++it) {
        const int cur = it & 1;
        const float *a_base = smem + cur * 384 * PITCH + (wr * 64 + l31) * PITCH + 4 * h;
        const float *b_base = smem + cur * 384 * PITCH + (256 + wc * 64 + l31) * PITCH + 4 * h;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            v4f a[2], b[2];
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const v4f *>(a_base + i * 32 * PITCH + kk * 8);
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const v4f *>(b_base + j * 32 * PITCH + kk * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        }
        if (mode == 0) {
            float *dst = smem + (cur ^ 1) * 384 * PITCH;
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<v4f *>(dst + (rsub + 64 * j) * PITCH + piece) = stage[j];
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    sink[blockIdx.x * 512 + tid] = s;
}

}  // namespace

extern "C" {

NSG_API int nsg_debug_lds_dma_loop(int32_t blocks, int32_t chunks, const float *src, float *sink, void *stream)
{
    NSG_REQUIRE(blocks > 0 && chunks > 0 && src && sink, NSG_E_INVALID, "nsg_debug_lds_dma_loop: bad argument");
    const size_t lds = (size_t)2 * 256 * 32 * sizeof(float);
    hipLaunchKernelGGL(lds_dma_loop_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, chunks, src, sink);
    return nsg_check_launch("lds_dma_loop_kernel");
}

NSG_API int nsg_debug_lds_fed_loop8(int32_t blocks, int32_t chunks, int32_t mode, float *sink, void *stream)
{
    NSG_REQUIRE(blocks > 0 && chunks > 0 && sink, NSG_E_INVALID, "nsg_debug_lds_fed_loop8: bad argument");
    const size_t lds = (size_t)2 * 384 * 36 * sizeof(float);
    static LdsOptIn once;
    if (const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&lds_fed_loop8_kernel)}, lds, "nsg_debug_lds_fed_loop8")) return rc;
    hipLaunchKernelGGL(lds_fed_loop8_kernel, dim3(blocks), dim3(512), lds, (hipStream_t)stream, chunks, mode, sink);
    return nsg_check_launch("lds_fed_loop8_kernel");
}

// blocks workgroups x chunks chunks of 128 x 128 x 64 (2 * 128 * 128 * 64 flop each); sink: blocks * 256 floats
NSG_API int nsg_debug_lds_fed_loop(int32_t blocks, int32_t chunks, int32_t mode, float *sink, void *stream)
{
    NSG_REQUIRE(blocks > 0 && chunks > 0 && sink, NSG_E_INVALID, "nsg_debug_lds_fed_loop: bad argument");
    const size_t lds = (size_t)2 * 256 * 36 * sizeof(float);
    static LdsOptIn once;
    if (const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&lds_fed_loop_kernel)}, lds, "nsg_debug_lds_fed_loop")) return rc;
    hipLaunchKernelGGL(lds_fed_loop_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, chunks, mode, sink);
    return nsg_check_launch("lds_fed_loop_kernel");
}


// sink: blocks*256 floats, stamps: blocks*2 u64 (shader-cycle delta, 100 MHz-tick delta per block)
NSG_API int nsg_debug_mfma_peak(int32_t blocks, int32_t iters, float *sink, unsigned long long *stamps, void *stream)
{
    NSG_REQUIRE(blocks > 0 && iters > 0 && sink && stamps, NSG_E_INVALID, "nsg_debug_mfma_peak: bad argument");
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, 1.0f, sink, stamps);
    return nsg_check_launch("mfma_peak_kernel");
}

}  // extern "C"
