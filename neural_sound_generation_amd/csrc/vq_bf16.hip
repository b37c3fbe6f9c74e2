// The quantiser's nearest-code search for the bf16 compute mode: the same fused ||x||^2 + ||e||^2 - 2 x.e^T + argmin +
// gather as vq.hip (reference: src/vector_quantization.py:6-23,40-42), with the contraction on the bf16 matrix pipe.
// Both fp32 operands are split into bf16 hi + lo parts (x = hi + lo + O(2^-17 |x|)) and the dot product is taken as
//   x.e ~= lo_x.hi_e + hi_x.lo_e + hi_x.hi_e      (fp32 accumulate; the dropped lo.lo term is O(2^-16) relative),
// i.e. three v_mfma_f32_32x32x16_bf16 per 16 channels instead of eight v_mfma_f32_32x32x2_f32: 5x less matrix-pipe
// time.  Distances carry a relative error of ~2^-16 of |x||e|, so this search is NOT the bit-exact one: on near-ties
// it may pick a different (equally near) code than the reference.  It is used only in the bf16 mode, whose encoder
// output already carries bf16 rounding (2^-9); the fp32 parity mode always runs vq.hip.
#include "nsg_common.h"
#include <math.h>

namespace {

// src [R][D] fp32 -> hi, lo [Rp][DP] bf16 (zero beyond R rows / D columns)
__global__ void split_bf16_kernel(const float *__restrict__ src, int R, int D, int Rp, int DP, bf16_t *__restrict__ hi,
                                  bf16_t *__restrict__ lo)
{
    const int64_t total = (int64_t)Rp * DP;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / DP), d = (int)(i - (int64_t)r * DP);
        const float v = (r < R && d < D) ? src[(size_t)r * D + d] : 0.f;
        const bf16_t h = nsg_f2bf(v);
        hi[i] = h;
        lo[i] = nsg_f2bf(v - nsg_bf2f(h));
    }
}

__device__ __forceinline__ void split8(const v4f a, const v4f b, bf16x8 &hi, bf16x8 &lo)
{
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    s16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bf16_t hb = nsg_f2bf(v[i]);
        h[i] = (short)hb;
        l[i] = (short)nsg_f2bf(v[i] - nsg_bf2f(hb));
    }
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

// Block = 4 waves = 128 rows of x (32 per wave, held as MFMA A fragments hi/lo); the pre-split codebook streams through
// double-buffered LDS in tiles of 32 codes; the (N, K) matrix is never materialised.
// HAS_X2 = false: no |x|^2 term (it changes no argmin; only the reported distances need it): 16 registers fewer, which at
// D = 128 is the difference between two and three blocks per CU (the kernel alternates between an HBM phase -- its rows in, the
// picked codes out -- and an MFMA phase, and only other blocks on the CU overlap the two)
template <int NKS, bool HAS_X2>   // k-steps of 16 channels: DP = 16 * NKS >= D
__global__ __launch_bounds__(256) void vq_forward_bf16x3_kernel(const float *__restrict__ x, const bf16_t *__restrict__ ehi,
                                                                const bf16_t *__restrict__ elo, const float *__restrict__ e,
                                                                const float *__restrict__ x2, const float *__restrict__ c2,
                                                                int64_t N, int D, int K, int64_t *__restrict__ idx_out,
                                                                float *__restrict__ codes_out, float *__restrict__ dmin_out,
                                                                bf16_t *__restrict__ codes_lp_out, int lp_relu,
                                                                const float *__restrict__ clip_rows, int64_t rows_per_clip)
{
    constexpr int DP = 16 * NKS;
    constexpr int EPB = DP * 2 + 16;              // LDS row pitch in bytes: 16 rows -> 16 different 16-byte slots
    constexpr int PLANE = 32 * EPB;               // one plane (hi or lo) of a 32-code tile
    constexpr int PIECES = 2 * 32 * DP / 8;       // 16-byte pieces per tile (both planes)
    constexpr int EJ = (PIECES + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][2][32][EPB]
    __shared__ int sidx[128];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128;

    // ---- A fragments: lane (l31, h) holds channels 16 s + 8 h .. + 7 of row row0 + 32 wave + l31 ----
    bf16x8 ahi[NKS], alo[NKS];
    {
        const int64_t row = row0 + wave * 32 + l31;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const int d0 = 16 * s + 8 * h;
            v4f a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (row < N && d0 < D) {               // D % 8 == 0: the 8 channels are all inside the row
                a = *reinterpret_cast<const v4f *>(x + row * D + d0);
                b = *reinterpret_cast<const v4f *>(x + row * D + d0 + 4);
            }
            split8(a, b, ahi[s], alo[s]);
        }
    }
    float x2v[HAS_X2 ? 16 : 1];
    if (HAS_X2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            x2v[r] = row < N ? x2[row] : 0.f;
        }
    }
    float best[16];
    int bidx[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { best[r] = INFINITY; bidx[r] = 0x7fffffff; }

    const int ntiles = (K + 31) / 32;             // ehi / elo hold 32 * ntiles rows (zero rows past K)
    v4f re[EJ];
    auto gload = [&](int ct) {
#pragma unroll
        for (int j = 0; j < EJ; ++j) {
            const int f = tid + 256 * j;
            if (PIECES % 256 != 0 && f >= PIECES) continue;
            const int plane = f / (PIECES / 2);
            const int rem = f - plane * (PIECES / 2);
            const int cr = rem / (DP / 8), c8 = (rem - cr * (DP / 8)) * 8;
            const bf16_t *src = (plane ? elo : ehi) + (size_t)(ct * 32 + cr) * DP + c8;
            re[j] = *reinterpret_cast<const v4f *>(src);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < EJ; ++j) {
            const int f = tid + 256 * j;
            if (PIECES % 256 != 0 && f >= PIECES) continue;
            const int plane = f / (PIECES / 2);
            const int rem = f - plane * (PIECES / 2);
            const int cr = rem / (DP / 8), c8 = (rem - cr * (DP / 8)) * 8;
            *reinterpret_cast<v4f *>(smem + (buf * 2 + plane) * PLANE + cr * EPB + c8 * 2) = re[j];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int ct = 0; ct < ntiles; ++ct) {
        const int cur = ct & 1;
        if (ct + 1 < ntiles) gload(ct + 1);
        const int code = ct * 32 + l31;
        const float c2v = code < K ? c2[code] : INFINITY;
        const char *bh = smem + (cur * 2 + 0) * PLANE + l31 * EPB + 16 * h;
        const char *bl = smem + (cur * 2 + 1) * PLANE + l31 * EPB + 16 * h;
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const bf16x8 eh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4f *>(bh + 32 * s));
            const bf16x8 el = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4f *>(bl + 32 * s));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[s], eh, acc, 0, 0, 0);   // small terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[s], el, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[s], eh, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dist = HAS_X2 ? __fmaf_rn(-2.0f, acc[r], __fadd_rn(c2v, x2v[r])) : __fmaf_rn(-2.0f, acc[r], c2v);
            if (dist < best[r]) { best[r] = dist; bidx[r] = code; }
        }
        if (ct + 1 < ntiles) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- first minimum across the 32 lanes (codes) of each half; xor < 32 stays inside the half ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float bv = best[r];
        int bi = bidx[r];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (bi == 0x7fffffff) bi = 0;
        if (l31 == 0) {
            const int rl = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int64_t row = row0 + rl;
            sidx[rl] = bi;
            if (row < N) {
                idx_out[row] = (int64_t)bi;
                if (dmin_out) dmin_out[row] = bv;
            }
        }
    }
    if (codes_out == nullptr && codes_lp_out == nullptr) return;
    __syncthreads();
    // ---- gather the (fp32) code rows: codes_out[row] = e[idx[row]] ----
    if (codes_out) {
        const int D4 = D >> 2;
        for (int f = tid; f < 128 * D4; f += 256) {
            const int r = f / D4, d4 = (f - r * D4) * 4;
            const int64_t row = row0 + r;
            if (row < N) *reinterpret_cast<v4f *>(codes_out + row * D + d4) = *reinterpret_cast<const v4f *>(e + (size_t)sidx[r] * D + d4);
        }
    }
    // ---- and / or their bf16 copy, optionally ReLU'd: what the decoder's leading ReLU consumes in the bf16 mode ----
    if (codes_lp_out) {
        const int D8 = D >> 3;
        for (int f = tid; f < 128 * D8; f += 256) {
            const int r = f / D8, d8 = (f - r * D8) * 8;
            const int64_t row = row0 + r;
            if (row >= N) continue;
            const float *src = e + (size_t)sidx[r] * D + d8;
            const v4f a = *reinterpret_cast<const v4f *>(src), b = *reinterpret_cast<const v4f *>(src + 4);
            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            if (clip_rows) {    // per-clip conditioning row (speaker embedding) added to every latent row of the clip, before the ReLU
                const float *cr = clip_rows + (size_t)(row / rows_per_clip) * D + d8;
                const v4f ca = *reinterpret_cast<const v4f *>(cr), cb = *reinterpret_cast<const v4f *>(cr + 4);
                v[0] += ca.x; v[1] += ca.y; v[2] += ca.z; v[3] += ca.w; v[4] += cb.x; v[5] += cb.y; v[6] += cb.z; v[7] += cb.w;
            }
            if (lp_relu) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            Elem<bf16_t>::store16(codes_lp_out + row * D + d8, v);
        }
    }
}

template <int NKS>
int launch(const float *x, const bf16_t *ehi, const bf16_t *elo, const float *e, const float *x2, const float *c2, int64_t N, int D,
           int K, int64_t *idx, float *codes, float *dmin, bf16_t *codes_lp, int lp_relu, const float *clip_rows, int64_t rows_per_clip,
           hipStream_t s)
{
    constexpr int DP = 16 * NKS;
    const size_t lds = (size_t)2 * 2 * 32 * (DP * 2 + 16);
    const int64_t nb = nsg_cdiv(N, 128);
    if (nb > 0x7fffffff) return nsg_fail(NSG_E_UNSUPPORTED, "vq_forward_bf16x3: too many rows");
    static LdsOptIn once;
    if (lds > 65536 - 1024) {
        const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&vq_forward_bf16x3_kernel<NKS, true>),
                                             reinterpret_cast<const void *>(&vq_forward_bf16x3_kernel<NKS, false>)}, lds, "vq_forward_bf16x3");
        if (rc != NSG_OK) return rc;
    }
    if (x2) hipLaunchKernelGGL((vq_forward_bf16x3_kernel<NKS, true>), dim3((unsigned)nb), dim3(256), lds, s, x, ehi, elo, e, x2, c2, N, D, K, idx, codes, dmin, codes_lp, lp_relu, clip_rows, rows_per_clip);
    else    hipLaunchKernelGGL((vq_forward_bf16x3_kernel<NKS, false>), dim3((unsigned)nb), dim3(256), lds, s, x, ehi, elo, e, x2, c2, N, D, K, idx, codes, dmin, codes_lp, lp_relu, clip_rows, rows_per_clip);
    return nsg_check_launch("vq_forward_bf16x3_kernel");
}

inline int dp_of(int D) { return D <= 16 ? 16 : D <= 32 ? 32 : D <= 64 ? 64 : D <= 128 ? 128 : 256; }
inline size_t split_bytes(int D, int K) { return nsg_align_up((size_t)nsg_cdiv(K, 32) * 32 * dp_of(D) * sizeof(bf16_t), 256); }

}  // namespace

extern "C" {

size_t nsg_vq_bf16x3_workspace_bytes(int64_t N, int32_t D, int32_t K)
{
    if (N < 0 || K <= 0 || D <= 0) return 0;
    return nsg_vq_workspace_bytes(N, D, K) + 2 * split_bytes(D, K);
}

int nsg_vq_forward_bf16x3(const float *x, const float *e, int64_t N, int32_t D, int32_t K, int64_t *idx_out, float *codes_out,
                          float *dmin_out, void *codes_bf16_out, int32_t bf16_relu, void *workspace, size_t workspace_bytes, void *stream)
{
    return nsg_vq_forward_bf16x3_cond(x, e, N, D, K, idx_out, codes_out, dmin_out, codes_bf16_out, bf16_relu, nullptr, 0, workspace,
                                      workspace_bytes, stream);
}

int nsg_vq_forward_bf16x3_cond(const float *x, const float *e, int64_t N, int32_t D, int32_t K, int64_t *idx_out, float *codes_out,
                               float *dmin_out, void *codes_bf16_out, int32_t bf16_relu, const float *clip_rows, int64_t rows_per_clip,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && e && idx_out && N >= 0 && D > 0 && K > 0, NSG_E_INVALID, "nsg_vq_forward_bf16x3: bad argument");
    NSG_REQUIRE(!clip_rows || (codes_bf16_out && rows_per_clip > 0 && nsg_aligned16(clip_rows)), NSG_E_INVALID,
                "nsg_vq_forward_bf16x3_cond: clip_rows needs codes_bf16_out, rows_per_clip > 0 and 16-byte alignment");
    NSG_REQUIRE(D <= 256 && D % 8 == 0, NSG_E_UNSUPPORTED, "nsg_vq_forward_bf16x3: D=%d must be a multiple of 8, at most 256", D);
    NSG_REQUIRE(nsg_aligned16(x) && nsg_aligned16(e) && (!codes_out || nsg_aligned16(codes_out)) && (!codes_bf16_out || nsg_aligned16(codes_bf16_out)),
                NSG_E_INVALID, "nsg_vq_forward_bf16x3: pointers must be 16-byte aligned");
    bf16_t *lp = reinterpret_cast<bf16_t *>(codes_bf16_out);
    if (N == 0) return NSG_OK;
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_vq_bf16x3_workspace_bytes(N, D, K), NSG_E_WORKSPACE,
                "nsg_vq_forward_bf16x3: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace);
    float *x2 = reinterpret_cast<float *>(ws);
    float *c2 = reinterpret_cast<float *>(ws + nsg_align_up((size_t)N * sizeof(float), 256));
    bf16_t *ehi = reinterpret_cast<bf16_t *>(ws + nsg_vq_workspace_bytes(N, D, K));
    bf16_t *elo = reinterpret_cast<bf16_t *>(ws + nsg_vq_workspace_bytes(N, D, K) + split_bytes(D, K));
    // |x|^2 is constant along a row: it only matters for the reported distances.  Without dmin_out it is neither computed
    // (a pass over x) nor added (the sum c2 + x2 would round the small code norms away)
    int rc = NSG_OK;
    if (dmin_out) rc = nsg_rowsumsq(x, N, D, x2, stream);
    else x2 = nullptr;
    if (rc) return rc;
    rc = nsg_rowsumsq(e, K, D, c2, stream);
    if (rc) return rc;
    const int DP = dp_of(D), Kp = (int)nsg_cdiv(K, 32) * 32;
    const int64_t tot = (int64_t)Kp * DP;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)(nsg_cdiv(tot, 256) > 1024 ? 1024 : nsg_cdiv(tot, 256))), dim3(256), 0, s, e, K, D,
                       Kp, DP, ehi, elo);
    rc = nsg_check_launch("split_bf16_kernel");
    if (rc) return rc;
    switch (DP) {
    case 16:  return launch<1>(x, ehi, elo, e, x2, c2, N, D, K, idx_out, codes_out, dmin_out, lp, bf16_relu, clip_rows, rows_per_clip, s);
    case 32:  return launch<2>(x, ehi, elo, e, x2, c2, N, D, K, idx_out, codes_out, dmin_out, lp, bf16_relu, clip_rows, rows_per_clip, s);
    case 64:  return launch<4>(x, ehi, elo, e, x2, c2, N, D, K, idx_out, codes_out, dmin_out, lp, bf16_relu, clip_rows, rows_per_clip, s);
    case 128: return launch<8>(x, ehi, elo, e, x2, c2, N, D, K, idx_out, codes_out, dmin_out, lp, bf16_relu, clip_rows, rows_per_clip, s);
    default:  return launch<16>(x, ehi, elo, e, x2, c2, N, D, K, idx_out, codes_out, dmin_out, lp, bf16_relu, clip_rows, rows_per_clip, s);
    }
}

}  // extern "C"
