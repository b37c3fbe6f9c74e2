// Internal helpers shared by the kernels of libnsg.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include <initializer_list>
#include "nsg.h"

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;   // storage type of a bfloat16 element (bit pattern)

// Bits of a float passed BY VALUE.  (__builtin_bit_cast applied directly to an ext-vector element
// expression such as `v.y` reads element 0 with this clang -- always go through these helpers.)
__host__ __device__ __forceinline__ unsigned nsg_fbits(float f) { return __builtin_bit_cast(unsigned, f); }
__host__ __device__ __forceinline__ float nsg_bitsf(unsigned u) { return __builtin_bit_cast(float, u); }

// fp32 -> bf16 bits, round to nearest even (v_cvt_pk_bf16_f32); bf16 bits -> fp32 is a 16-bit shift
__device__ __forceinline__ bf16_t nsg_f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// two at once: ONE v_cvt_pk_bf16_f32 (lo = a, hi = b).  (Converting singly and packing by hand costs three instructions per pair.)
typedef float nsg_v2f __attribute__((ext_vector_type(2)));
typedef __bf16 nsg_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned nsg_pack_bf16(float a, float b)
{
    const nsg_v2f f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, nsg_bf16x2));
}
__device__ __forceinline__ float nsg_bf2f(bf16_t h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// Elem<T>: N = elements per 16-byte access; load16 / store16 move N elements as floats
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack16(const v4f v, float *o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    static __device__ __forceinline__ void load16(const float *p, float *o) { unpack16(*reinterpret_cast<const v4f *>(p), o); }
    static __device__ __forceinline__ void store16(float *p, const float *o) { v4f v = {o[0], o[1], o[2], o[3]}; *reinterpret_cast<v4f *>(p) = v; }
    static __device__ __forceinline__ float get(const float *p) { return *p; }
    static __device__ __forceinline__ void put(float *p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load16(const bf16_t *p, float *o) { unpack16(*reinterpret_cast<const v4f *>(p), o); }
    static __device__ __forceinline__ void unpack16(const v4f raw, float *o)
    {
        const float r0 = raw.x, r1 = raw.y, r2 = raw.z, r3 = raw.w;
        const unsigned u[4] = {nsg_fbits(r0), nsg_fbits(r1), nsg_fbits(r2), nsg_fbits(r3)};
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = nsg_bitsf(u[i] << 16); o[2 * i + 1] = nsg_bitsf(u[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store16(bf16_t *p, const float *o)
    {
        unsigned u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = nsg_pack_bf16(o[2 * i], o[2 * i + 1]);
        v4f raw = {nsg_bitsf(u[0]), nsg_bitsf(u[1]), nsg_bitsf(u[2]), nsg_bitsf(u[3])};
        *reinterpret_cast<v4f *>(p) = raw;
    }
    static __device__ __forceinline__ float get(const bf16_t *p) { return nsg_bf2f(*p); }
    static __device__ __forceinline__ void put(bf16_t *p, float v) { *p = nsg_f2bf(v); }
};

// Fixed-order sum of n floats p[0], p[stride], p[2*stride], ... : the loads are issued 16 at a time (the
// finalize kernels are otherwise a chain of dependent global-load latencies), the adds stay in index order.
template <typename ACC>
__device__ __forceinline__ ACC nsg_strided_sum(const float *__restrict__ p, size_t stride, int n)
{
    ACC s = (ACC)0;
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = p[(size_t)(i + k) * stride];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += (ACC)v[k];
    }
    if (i + 8 <= n) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(i + k) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (ACC)v[k];
        i += 8;
    }
    for (; i < n; ++i) s += (ACC)p[(size_t)i * stride];
    return s;
}

// two arrays at once (their loads share the latency)
__device__ __forceinline__ void nsg_strided_sum2(const float *__restrict__ p, const float *__restrict__ q, size_t stride, int n,
                                                 double &sp, double &sq)
{
    double a = 0.0, b = 0.0;
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        float v[16], w[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[k] = p[(size_t)(i + k) * stride]; w[k] = q[(size_t)(i + k) * stride]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) { a += (double)v[k]; b += (double)w[k]; }
    }
    for (; i < n; ++i) { a += (double)p[(size_t)i * stride]; b += (double)q[(size_t)i * stride]; }
    sp = a;
    sq = b;
}

// thread-local error text (never thrown across the ABI)
void nsg_set_error(const char *fmt, ...);
int nsg_fail(int code, const char *fmt, ...);
int nsg_check_launch(const char *what);

#define NSG_REQUIRE(cond, code, ...)                    \
    do {                                                \
        if (!(cond)) return nsg_fail((code), __VA_ARGS__); \
    } while (0)

// More than 64 KiB of dynamic LDS must be opted into per kernel AND per device.  One LdsOptIn per call site (a function-local
// static: constant-initialised, so thread-safe) remembers the devices already done as a bit mask; racing first calls both
// set the (idempotent) attribute.  No lock, no other global state.
struct LdsOptIn {
    std::atomic<uint64_t> devices{0};
};
int nsg_lds_opt_in(LdsOptIn &once, std::initializer_list<const void *> kernels, size_t lds_bytes, const char *what);

// Run-time switches between kernel variants exist only in the diagnostics library (libnsg_diag.so, built with -DNSG_DIAG for
// scripts/ and the A/B tests): there they are file-static ints behind nsg_debug_set_* entry points; in libnsg.so they are
// compile-time constants (no mutable global state, no undeclared exports).
#ifdef NSG_DIAG
#define NSG_DIAG_SWITCH(type, name, init) static type name = init;
#else
#define NSG_DIAG_SWITCH(type, name, init) static constexpr type name = init;
#endif

static inline bool nsg_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__device__ __forceinline__ bool nsg_aligned16_dev(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t nsg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t nsg_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// W consecutive elements <-> floats, W a multiple of the type's 16-byte element count
template <typename T, int W>
__device__ __forceinline__ void ldw(const T *p, float *o)
{
#pragma unroll
    for (int k = 0; k < W; k += Elem<T>::N) Elem<T>::load16(p + k, o + k);
}
template <typename T, int W>
__device__ __forceinline__ void stw(T *p, const float *o)
{
#pragma unroll
    for (int k = 0; k < W; k += Elem<T>::N) Elem<T>::store16(p + k, o + k);
}
template <typename A, typename B>
struct Width { static constexpr int W = (sizeof(A) == 2 || sizeof(B) == 2) ? 8 : 4; };

// Division of n < 2^31 by a launch-invariant d >= 1 as one 64-bit multiply + shift (exact: with
// s = ceil(log2 d) and M = ceil(2^(31+s) / d) the error term n*(M*d - 2^(31+s)) stays below 2^(31+s)).
struct FastDiv {
    uint32_t mul;
    uint32_t shift;
};
static inline FastDiv nsg_fastdiv(uint32_t d)
{
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    const unsigned __int128 num = (unsigned __int128)1 << (31 + s);
    FastDiv f;
    f.mul = (uint32_t)((num + d - 1) / d);
    f.shift = 31 + s;
    return f;
}
__device__ __forceinline__ int nsg_div(int n, FastDiv f)
{
    return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.shift);
}

// ---- implicit-GEMM parameter blocks (gemm_gather.hip / gemm_wgrad.hip) -------------------------
struct GatherGemmParams {
    const void *in;     // [B][IH][IW][CI]      elements of in_dtype
    const void *w;      // [tap image][CO][CI]  elements of in_dtype
    const float *bias;  // [CO] fp32 or null
    void *out;          // [B][OH][OW][CO]      elements of out_dtype
    int in_dtype, out_dtype;   // NSG_F32 / NSG_BF16
    int B, IH, IW, CI;
    int OH, OW, CO;
    int KH, KW, stride, pad, pad_w;   // pad = rows, pad_w = columns
    int mode;    // 0: conv gather (iy = ry*stride - pad + kh); 1: transposed 4/2/1, one parity class per blockIdx.y
    int M;       // rows per class = B*RH*RW
    int RH, RW;  // row grid: conv -> (OH,OW); transposed -> (ceil(OH/2), ceil(OW/2))
    int flags;   // NSG_RELU_IN | NSG_TANH_OUT
    const void *epi_add;   // optional, laid out like `out` (elements of out_dtype): out = acc + bias + epi_add ...
    const void *epi_mask;  // optional, laid out like `out`: ... then zeroed where epi_mask <= 0 (gradient through a ReLU)
    FastDiv div_rw, div_rhw;     // division by RW and by RH*RW (filled in by nsg_launch_gather_gemm)
    unsigned in_bytes, w_bytes;  // sizes of `in` and `w` (filled in by nsg_launch_gather_gemm; buffer-load range checks)
    unsigned long long *stamps;  // diagnostics only: per block (shader cycles, 100 MHz ticks) spent in the main loop
    float *stats;  // optional [n_row_tiles][3][CO]: per row tile (valid-row count, mean, M2 about it) of the OUTPUT
};
// number of 128-row tiles (x parity classes) a launch with these parameters produces = rows of `stats`
int nsg_gather_gemm_row_tiles(const GatherGemmParams &p);

struct WgradParams {
    const void *P;       // [B][PH][PW][A]  tensor on the conv-OUTPUT pixel grid (rows of the reduction), elements of dtype
    const void *Q;       // [B][QH][QW][C]  tensor on the conv-INPUT pixel grid, gathered per tap, elements of dtype
    int dtype;           // NSG_F32 / NSG_BF16 (bf16 operands are widened to fp32 while staged: fp32 accumulate AND fp32 MFMA)
    const int64_t *idx;  // one-hot mode: P[pix][a] = (idx[pix] == a)
    float *partial;      // [nslab][ntaps][A][C]
    int B, PH, PW, A;
    int QH, QW, C;
    int KH, KW, stride, pad, pad_w;   // pad = rows, pad_w = columns
    int Mp;         // B*PH*PW
    int slab_rows;  // multiple of 32
    int relu_p, relu_q, onehot;
    // bf16 1x1 only, or null: Q is used as max(fma(Q, fs, off), 0), fs = invstd*gamma, off = beta - mean*fs (BatchNorm + ReLU on the operand's way in)
    const float *q_mean, *q_invstd, *q_gamma, *q_beta;
    FastDiv div_pw, div_phw;   // filled in by nsg_launch_wgrad
    unsigned p_bytes, q_bytes; // sizes of P and Q in bytes (filled in by nsg_launch_wgrad; buffer-load range checks)
    unsigned long long *stamps;  // diagnostics only
    int stagger;                 // s_sleep units (64 cycles) the second-dispatched half of the grid waits at start
};

// A packed bf16 weight image [taps][NN][CC] is followed by its fragment-ordered twin (conv_api.hip: PackJob::frag) when:
static inline bool nsg_frag_image(int NN, int CC) { return NN % 128 == 0 && CC % 64 == 0; }
// Order of that twin = v_mfma_f32_32x32x16_bf16's operand fragments as gemm_patch.hip's waves load them:
//   [t][c/64][n/32][(c%64)/16][lane = n%32 + 32*((c%16)/8)][c%8]
// (a v_mfma_f32_16x16x32_bf16 order was measured at parity inside the step, 955-961 vs 946 TF, and removed)
__host__ __device__ static inline int64_t nsg_frag_index(int t, int n, int c, int NN, int CC)
{
    const int64_t blk = ((int64_t)(t * (CC >> 6) + (c >> 6)) * (NN >> 5) + (n >> 5)) * 4;
    return ((blk + ((c & 63) >> 4)) * 64 + (n & 31) + 32 * ((c & 15) >> 3)) * 8 + (c & 7);
}
int nsg_launch_gather_gemm(const GatherGemmParams &p, hipStream_t s);

// ---- nearest-code search: codebook slices (vq.hip: the exact fp32 search) --------------------------------------------------------------
// A search block holds 128 rows and walks the whole codebook; two blocks share a CU.  When the row blocks alone do not fill
// the resident slots evenly (BASELINE configs[3]: 640 blocks on 512 slots = two rounds for 1.25 rounds of work), the codebook is
// cut into S slices (blockIdx.y), each block reports its slice's first minimum per row, and a combine pass takes the first
// minimum over the slices in slice order -- the same (distance, index) the unsplit search returns, bit for bit -- and gathers.
constexpr int NSG_SEARCH_CUS = 256;                         // CUs of an MI355X (speed only): two search blocks per CU up to D = 128, one beyond
int nsg_vq_slices(int64_t N, int D, int K);                 // S: 1, 2, 4 or 8
size_t nsg_vq_slice_bytes(int64_t N, int D, int K);         // [S][N] floats + [S][N] ints behind the searches' other workspace; 0 for S = 1
// idx / dmin / codes / bf16 codes (each optional except idx) from the S partial results; clip_rows as in nsg_vq_forward_bf16x3_cond
int nsg_launch_vq_combine(const float *pd, const int *pi, int S, int64_t N, int D, int K, const float *e, int64_t *idx, float *codes,
                          float *dmin, bf16_t *codes_lp, int lp_relu, const float *clip_rows, int64_t rows_per_clip, hipStream_t s);
// gemm_patch.hip: the patch-staged bf16 kernel for the shapes it implements (3x3/1, 4x4/2 and the transposed 4/2/1 with
// C_in % 64 == 0, C_out % 128 == 0, bf16 in and out); p must have in_bytes / w_bytes filled in.  *handled = false -> not run.
int nsg_launch_patch_gemm(const GatherGemmParams &p, hipStream_t s, bool *handled);
int nsg_patch_gemm_stat_records(const GatherGemmParams &p);     // statistics records (p.stats) such a launch writes; 0: not taken
// Returns bytes of partial-slab workspace it will use for these sizes.
size_t nsg_wgrad_workspace_bytes(int64_t Mp, int ntaps, int A, int C);
// dst[(a*C + c)*ntaps + t] = sum over slabs (fixed order) of partial; dst fully overwritten.
int nsg_launch_wgrad(WgradParams p, float *dst, void *ws, size_t ws_bytes, hipStream_t s);
int nsg_launch_wgrad_reduce(const float *partial, float *dst, int nslab, int ntaps, int A, int C, hipStream_t s);

// tile-statistics records [ntiles][3][C] (count, sum, M2 about the tile mean) -> mean / invstd / running statistics.
// The buffer must be nsg_bn_tiles_bytes(ntiles, C) long (the finalizer keeps its chunk records behind the tiles).
size_t nsg_bn_tiles_bytes(int64_t ntiles, int C);
int nsg_bn_stats_from_tiles(const float *tiles, int ntiles, int64_t M, int C, float eps, float momentum, float *mean,
                            float *invstd, float *running_mean, float *running_var, hipStream_t s);
// dbeta[c] = sum_s partial[s][0][c], dgamma[c] = sum_s partial[s][1][c] over nslab <= 1024 slabs of [2][C] (fixed order, double)
void nsg_bn_slab_geom(int64_t M, int *nslab, int *rows);      // bn.hip: slabs of nsg_bn_backward_sums over M rows
int nsg_launch_bn_bwd_final(const float *partial, int nslab, int C, float *dgamma, float *dbeta, hipStream_t s);
// ... and colsum[c] = sum_s colsum_partial[s][c] in the same launch
int nsg_launch_bn_bwd_final_wreduce(const float *partial, const float *colsum_partial, int nslab, int C, float *dgamma, float *dbeta, float *colsum,
                                    const float *wpartial, float *wdst, int wn, hipStream_t s);
int nsg_launch_bn_bwd_final_colsum(const float *partial, const float *colsum_partial, int nslab, int C, float *dgamma, float *dbeta,
                                   float *colsum, hipStream_t s);
// out[c] = sum_s partial[s][c] over nslab <= 1024 slabs of [C] (fixed order, double)
int nsg_launch_slab_sum_final(const float *partial, int nslab, int C, float *out, hipStream_t s);
