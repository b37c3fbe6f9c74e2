// The fused input layer Conv2d(1, C, 4, 2, 1) -> BatchNorm2d -> ReLU (src/models.py:165-167; C ABI nsg_c1conv_bn_relu_*)
// in the throughput (bf16 storage) mode, with the convolution on the matrix cores.
//
// stencil_c1.hip's passes recompute the conv output h on the vector ALU: 16 multiply-adds per element, which makes every
// pass ALU-bound (190 us for the 335M elements of the headline shape even with no tensor traffic at all).  Here a
// 32-pixel x 32-channel block of h is ONE v_mfma_f32_32x32x16_bf16 -- the 16 taps are exactly its K -- issued three
// times on (hi, lo) bf16 splits of the fp32 patch values and weights (lo*hi + hi*lo + hi*hi, fp32 accumulate: h to
// ~2^-17 relative, far inside the bf16 storage of everything downstream), so the vector ALU only does the per-element
// BatchNorm arithmetic and the passes run at the speed of their one tensor stream.
//
// Layout: block = NW waves, wave w owns channels 32w .. 32w+31 (C = 32 NW <= 128).  A tile is TW = 64 pixels of one row
// = two pixel blocks.  MFMA operands: A = patch [pixel][tap], B = weights [tap][channel]; in the result a LANE IS A
// CHANNEL (column = lane & 31) and its 16 registers are pixels (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the block, so
//   * every per-channel constant is one register and every per-channel sum accumulates in-lane;
//   * the gradient block d [pixel][channel] is directly the B operand [K = pixel][N = channel] of the weight-gradient
//     MFMA  dw^T[tap][channel] += patch^T[tap][pixel] * d[pixel][channel]  (bf16 operands, as every weight gradient of
//     this mode);
//   * the tensor tile (dy in, y out) crosses between this layout and 16-byte global accesses through LDS.
// All four passes build h with the same instruction sequence, so the ReLU decision of the forward is reproduced exactly
// by the backward.  Expressions (per channel, fs = invstd*gamma, off = beta - mean*fs):
//   t = fma(h, fs, off);  y = max(t, 0);  g = t > 0 ? dy : 0;
//   sums:  sum g,  sum g*(h - mean)   (x invstd at the end);
//   dh = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) = fma(sc, g, -fma(k1, h, k0)),
//        sc = gamma*invstd, k1 = sc*invstd*dgamma/M, k0 = sc*dbeta/M - k1*mean.
#include "nsg_common.h"
#include "c1_geom.h"
#include <type_traits>

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

template <int NT> struct PatchRegs { static constexpr int N = (PATCH_VALS + NT - 1) / NT; };

template <int NT>
__device__ __forceinline__ void patch_load(float (&r)[PatchRegs<NT>::N], const float *__restrict__ img, const C1Geom &g, int b, int ly, int ox0, int tid)
{
    constexpr int NCOL = 2 * TW + 2;
#pragma unroll
    for (int q = 0; q < PatchRegs<NT>::N; ++q) {
        const int i = tid + NT * q;
        const int rr = i / NCOL, cix = i - rr * NCOL;
        const int y = 2 * ly - 1 + rr, x = 2 * ox0 - 1 + cix;
        const bool ok = (i < PATCH_VALS) & (y >= 0) & (y < g.HH) & (x >= 0) & (x < g.WW);
        const float v = img[ok ? ((size_t)b * g.HH + y) * g.WW + x : 0];     // clamped, unconditional
        r[q] = ok ? v : 0.f;
    }
}
template <int NT>
__device__ __forceinline__ void patch_store(float *patch, const float (&r)[PatchRegs<NT>::N], int tid)
{
    constexpr int NCOL = 2 * TW + 2;
#pragma unroll
    for (int q = 0; q < PatchRegs<NT>::N; ++q) {
        const int i = tid + NT * q;
        if (i < PATCH_VALS) {
            const int rr = i / NCOL, cix = i - rr * NCOL;
            patch[rr * PP + cix] = r[q];
        }
    }
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
// two floats -> two bf16 in one register (one v_cvt_pk_bf16_f32, round to nearest even): a in the low half
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const v2f f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 &hi, bf16x8 &lo)
{
    v4u h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned hp = pack_bf16(v[2 * i], v[2 * i + 1]);
        h[i] = hp;
        l[i] = pack_bf16(v[2 * i] - nsg_bitsf(hp << 16), v[2 * i + 1] - nsg_bitsf(hp & 0xffff0000u));
    }
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

// The wave's share of the conv weights as the MFMA B operand: lane (n, hb) holds taps 8hb .. 8hb+7 of channel c
__device__ __forceinline__ void load_w_operand(const float *__restrict__ w, int c, int hb, bf16x8 &whi, bf16x8 &wlo)
{
    const v4f a = *reinterpret_cast<const v4f *>(w + (size_t)c * 16 + 8 * hb);
    const v4f b = *reinterpret_cast<const v4f *>(w + (size_t)c * 16 + 8 * hb + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    split8(v, whi, wlo);
}

// h of pixel block mb (32 pixels) x the wave's 32 channels.  A operand: lane (m, hb) holds taps 8hb .. 8hb+7 of pixel
// j = 32 mb + m, i.e. patch rows 2hb, 2hb+1, columns 2j .. 2j+3.
__device__ __forceinline__ v16f conv_block(const float *patch, int mb, int lane, const bf16x8 whi, const bf16x8 wlo, float bias)
{
    const int m = lane & 31, hb = lane >> 5;
    const float *p0 = patch + (2 * hb) * PP + 2 * (32 * mb + m);
    const v2f a = *reinterpret_cast<const v2f *>(p0), b = *reinterpret_cast<const v2f *>(p0 + 2);
    const v2f c = *reinterpret_cast<const v2f *>(p0 + PP), d = *reinterpret_cast<const v2f *>(p0 + PP + 2);
    const float v[8] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
    bf16x8 ahi, alo;
    split8(v, ahi, alo);
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, whi, acc, 0, 0, 0);   // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, wlo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, whi, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += bias;
    return acc;
}

// pixel (within its 32-pixel block) of accumulator register r of a lane in half hb
__device__ __forceinline__ int acc_pixel(int r, int hb) { return (r & 3) + 8 * (r >> 2) + 4 * hb; }

constexpr int gpitch(int C) { return C + 8; }   // bf16 elements per LDS row of a tensor tile: rows 4 apart land 16 banks apart

// ---- tensor tile (64 pixels x C channels, bf16) between global memory (16-byte pieces) and LDS ----
// NT = 2C threads: 8C pieces of 16 bytes = 4 per thread.
template <int NW>
__device__ __forceinline__ void gtile_load(v4f (&r)[4], const bf16_t *__restrict__ t, const C1Geom &g, int b, int ly, int ox0, int npx, int tid)
{
    constexpr int C = 32 * NW, NT = 64 * NW, CPR = C / 8;    // pieces per row
    const bf16_t *base = t + (((size_t)b * g.LH + ly) * g.LW + ox0) * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + NT * i;
        const int row = q / CPR, cc = q - row * CPR;
        r[i] = *reinterpret_cast<const v4f *>(base + (size_t)(row < npx ? row : 0) * C + 8 * cc);   // clamped: always inside the tensor
    }
}
template <int NW>
__device__ __forceinline__ void gtile_to_lds(bf16_t *buf, const v4f (&r)[4], int tid)
{
    constexpr int C = 32 * NW, NT = 64 * NW, CPR = C / 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + NT * i;
        const int row = q / CPR, cc = q - row * CPR;
        *reinterpret_cast<v4f *>(buf + row * gpitch(C) + 8 * cc) = r[i];
    }
}
template <int NW>
__device__ __forceinline__ void gtile_store(bf16_t *__restrict__ t, const bf16_t *buf, const C1Geom &g, int b, int ly, int ox0, int npx, int tid)
{
    constexpr int C = 32 * NW, NT = 64 * NW, CPR = C / 8;
    bf16_t *base = t + (((size_t)b * g.LH + ly) * g.LW + ox0) * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + NT * i;
        const int row = q / CPR, cc = q - row * CPR;
        if (row < npx) *reinterpret_cast<v4f *>(base + (size_t)row * C + 8 * cc) = *reinterpret_cast<const v4f *>(buf + row * gpitch(C) + 8 * cc);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward, pass 1: tiles[block][3][C] = (count, sum, M2 about the block mean) of h
// ---------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW) void c1m_stats_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                             const float *__restrict__ bias, float *__restrict__ tiles, const C1Geom g)
{
    constexpr int NT = 64 * NW, C = 32 * NW;
    __shared__ __attribute__((aligned(16))) float patch[2][4 * PP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hb = lane >> 5, c = 32 * wave + n;
    bf16x8 whi, wlo;
    load_w_operand(w, c, hb, whi, wlo);
    const float bs = bias ? bias[c] : 0.f;
    float pv = 0.f, s1 = 0.f, s2 = 0.f, cnt = 0.f;

    int buf = 0;
    float pr[PatchRegs<NT>::N];
    if ((int)blockIdx.x < g.ntiles) {
        int b, ly, ox0;
        tile_coords(g, blockIdx.x, b, ly, ox0);
        patch_load<NT>(pr, img, g, b, ly, ox0, tid);
        patch_store<NT>(patch[0], pr, tid);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x, buf ^= 1) {
        int b, ly, ox0;
        tile_coords(g, tile, b, ly, ox0);
        const int nxt = tile + gridDim.x;
        if (nxt < g.ntiles) {
            int nb, nly, nox0;
            tile_coords(g, nxt, nb, nly, nox0);
            patch_load<NT>(pr, img, g, nb, nly, nox0, tid);
        }
        const int npx = min(TW, g.LW - ox0);
        if (npx == TW) {          // full tile (the common case): every register is a real pixel
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const v16f h = conv_block(patch[buf], mb, lane, whi, wlo, bs);
                pv = cnt == 0.f ? h[0] : pv;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = h[r] - pv;
                    s1 += d;
                    s2 = __builtin_fmaf(d, d, s2);
                }
                cnt += 16.f;
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if (32 * mb >= npx) break;
                const v16f h = conv_block(patch[buf], mb, lane, whi, wlo, bs);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = 32 * mb + acc_pixel(r, hb) < npx;
                    pv = (cnt == 0.f && ok) ? h[r] : pv;
                    const float d = ok ? h[r] - pv : 0.f;
                    s1 += d;
                    s2 = __builtin_fmaf(d, d, s2);
                    cnt += ok ? 1.f : 0.f;
                }
            }
        }
        if (nxt < g.ntiles) patch_store<NT>(patch[buf ^ 1], pr, tid);
        __syncthreads();
    }
    // the lane's (count, mean, M2), pooled with the other half-wave's lane of the same channel
    const double n0 = cnt, inv = cnt > 0.f ? 1.0 / cnt : 0.0;
    const double m0 = (double)pv + (double)s1 * inv;
    double q0 = (double)s2 - (double)s1 * (double)s1 * inv;
    q0 = q0 > 0.0 ? q0 : 0.0;
    const double n1 = __shfl_xor(n0, 32, 64), m1 = __shfl_xor(m0, 32, 64), q1 = __shfl_xor(q0, 32, 64);
    if (hb == 0) {
        const double N = n0 + n1;
        const double mu = N > 0.0 ? (n0 * m0 + n1 * m1) / N : 0.0;
        const double Q = q0 + q1 + n0 * (m0 - mu) * (m0 - mu) + n1 * (m1 - mu) * (m1 - mu);
        float *dst = tiles + (size_t)blockIdx.x * 3 * C;
        dst[c] = (float)N;
        dst[C + c] = (float)(N * mu);
        dst[2 * C + c] = (float)Q;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward, pass 2: y = max(fma(h, fs, off), 0) as bf16
// ---------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW) void c1m_apply_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                             const float *__restrict__ bias, const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, bf16_t *__restrict__ out, const C1Geom g)
{
    constexpr int NT = 64 * NW, C = 32 * NW;
    __shared__ __attribute__((aligned(16))) float patch[2][4 * PP];
    __shared__ __attribute__((aligned(16))) bf16_t ybuf[2][TW * gpitch(C)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hb = lane >> 5, c = 32 * wave + n;
    bf16x8 whi, wlo;
    load_w_operand(w, c, hb, whi, wlo);
    const float bs = bias ? bias[c] : 0.f;
    const float fs = invstd[c] * gamma[c];
    const float off = __builtin_fmaf(-mean[c], fs, beta[c]);

    int buf = 0;
    float pr[PatchRegs<NT>::N];
    if ((int)blockIdx.x < g.ntiles) {
        int b, ly, ox0;
        tile_coords(g, blockIdx.x, b, ly, ox0);
        patch_load<NT>(pr, img, g, b, ly, ox0, tid);
        patch_store<NT>(patch[0], pr, tid);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x, buf ^= 1) {
        int b, ly, ox0;
        tile_coords(g, tile, b, ly, ox0);
        const int nxt = tile + gridDim.x;
        if (nxt < g.ntiles) {
            int nb, nly, nox0;
            tile_coords(g, nxt, nb, nly, nox0);
            patch_load<NT>(pr, img, g, nb, nly, nox0, tid);
        }
        const int npx = min(TW, g.LW - ox0);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            if (32 * mb >= npx) break;
            const v16f h = conv_block(patch[buf], mb, lane, whi, wlo, bs);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y = fmaxf(__builtin_fmaf(h[r], fs, off), 0.f);
                ybuf[buf][(32 * mb + acc_pixel(r, hb)) * gpitch(C) + c] = nsg_f2bf(y);    // rows >= npx: inside the buffer, never stored
            }
        }
        if (nxt < g.ntiles) patch_store<NT>(patch[buf ^ 1], pr, tid);
        __syncthreads();
        gtile_store<NW>(out, ybuf[buf], g, b, ly, ox0, npx, tid);     // (the next tile writes the other buffer)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, pass 1: partial[block][2][C] = (sum g, sum g*xhat)
// SWAP = false: the input layer (BatchNorm input = the recomputed conv value h, incoming gradient = the tensor dy).
// SWAP = true:  the OUTPUT layer BatchNorm -> ReLU -> ConvTranspose2d(C, 1, 4, 2, 1) (src/models.py:180-182): the roles
//               are exchanged -- the tensor is the BatchNorm input u, and the gradient that reaches the ReLU is the data
//               gradient of the transposed conv, da[pix][c] = sum_t patch(dpre)[t] * w[c][t]: the same MFMA block, built
//               from the gradient image dpre and never stored.
// ---------------------------------------------------------------------------------------------------------------
template <int NW, bool SWAP>
__global__ __launch_bounds__(64 * NW) void c1m_bwd_sums_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                                const float *__restrict__ bias, const bf16_t *__restrict__ dy,
                                                                const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                float *__restrict__ partial, const C1Geom g)
{
    constexpr int NT = 64 * NW, C = 32 * NW;
    __shared__ __attribute__((aligned(16))) float patch[2][4 * PP];
    __shared__ __attribute__((aligned(16))) bf16_t gbuf[2][TW * gpitch(C)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hb = lane >> 5, c = 32 * wave + n;
    bf16x8 whi, wlo;
    load_w_operand(w, c, hb, whi, wlo);
    const float bs = bias ? bias[c] : 0.f;
    const float mu = mean[c], is = invstd[c];
    const float fs = is * gamma[c];
    const float off = __builtin_fmaf(-mu, fs, beta[c]);
    float s1 = 0.f, s2 = 0.f;

    int buf = 0;
    float pr[PatchRegs<NT>::N];
    v4f gr[4];
    if ((int)blockIdx.x < g.ntiles) {
        int b, ly, ox0;
        tile_coords(g, blockIdx.x, b, ly, ox0);
        patch_load<NT>(pr, img, g, b, ly, ox0, tid);
        gtile_load<NW>(gr, dy, g, b, ly, ox0, min(TW, g.LW - ox0), tid);
        patch_store<NT>(patch[0], pr, tid);
        gtile_to_lds<NW>(gbuf[0], gr, tid);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x, buf ^= 1) {
        int b, ly, ox0;
        tile_coords(g, tile, b, ly, ox0);
        const int nxt = tile + gridDim.x;
        if (nxt < g.ntiles) {
            int nb, nly, nox0;
            tile_coords(g, nxt, nb, nly, nox0);
            patch_load<NT>(pr, img, g, nb, nly, nox0, tid);
            gtile_load<NW>(gr, dy, g, nb, nly, nox0, min(TW, g.LW - nox0), tid);
        }
        const int npx = min(TW, g.LW - ox0);
        auto body = [&](auto FULL) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if (!decltype(FULL)::value && 32 * mb >= npx) break;
                const v16f h = conv_block(patch[buf], mb, lane, whi, wlo, bs);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = 32 * mb + acc_pixel(r, hb);
                    const float tv = nsg_bf2f(gbuf[buf][j * gpitch(C) + c]);
                    const float x = SWAP ? tv : h[r], gv = SWAP ? h[r] : tv;     // BatchNorm input, incoming gradient
                    bool pass = __builtin_fmaf(x, fs, off) > 0.f;
                    if (!decltype(FULL)::value) pass = pass & (j < npx);
                    const float ge = pass ? gv : 0.f;
                    s1 += ge;
                    s2 = __builtin_fmaf(ge, x - mu, s2);
                }
            }
        };
        if (npx == TW) body(std::true_type{}); else body(std::false_type{});
        if (nxt < g.ntiles) {
            patch_store<NT>(patch[buf ^ 1], pr, tid);
            gtile_to_lds<NW>(gbuf[buf ^ 1], gr, tid);
        }
        __syncthreads();
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (hb == 0) {
        float *dst = partial + (size_t)blockIdx.x * 2 * C;
        dst[c] = s1;
        dst[C + c] = s2 * is;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, pass 2: dh on the spot, folded into partial[block][c][17] = 16 tap sums of dh*patch + the column sum of dh
// ---------------------------------------------------------------------------------------------------------------
// SWAP = true (the output layer, see c1m_bwd_sums_kernel): the tensor is u; du = BatchNorm backward of the recomputed da is
// written back over the tile in LDS and stored (dx_out), and the weight gradient of the transposed conv accumulates
// dw^T[tap][c] += patch(dpre)^T[tap][pixel] * a[pixel][c] with a = max(fma(u, fs, off), 0) rebuilt on the spot.
template <int NW, bool SWAP>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3))) void c1m_bwd_wgrad_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                                 const float *__restrict__ bias, const bf16_t *__restrict__ dy,
                                                                 const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                                 float inv_m, float *__restrict__ partial, bf16_t *__restrict__ dx_out,
                                                                 const C1Geom g)
{
    constexpr int NT = 64 * NW, C = 32 * NW;
    __shared__ __attribute__((aligned(16))) float patch[2][4 * PP];
    __shared__ __attribute__((aligned(16))) bf16_t gbuf[2][TW * gpitch(C)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hb = lane >> 5, c = 32 * wave + n;
    bf16x8 whi, wlo;
    load_w_operand(w, c, hb, whi, wlo);
    const float bs = bias ? bias[c] : 0.f;
    const float mu = mean[c], is = invstd[c];
    const float fs = is * gamma[c];
    const float off = __builtin_fmaf(-mu, fs, beta[c]);
    const float sc = gamma[c] * is;
    const float k1 = sc * is * (dgamma[c] * inv_m);
    const float k0 = __builtin_fmaf(sc, dbeta[c] * inv_m, -(k1 * mu));
    // this lane as a row of the patch^T operand: tap n (rows 16..31 of the operand are zero)
    const unsigned tap_mask = n < 16 ? 0xffffffffu : 0u;
    const int tap_off = ((n >> 2) & 3) * PP + (n & 3);
    float cs = 0.f;
    v16f dwacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dwacc[r] = 0.f;

    int buf = 0;
    float pr[PatchRegs<NT>::N];
    v4f gr[4];
    if ((int)blockIdx.x < g.ntiles) {
        int b, ly, ox0;
        tile_coords(g, blockIdx.x, b, ly, ox0);
        patch_load<NT>(pr, img, g, b, ly, ox0, tid);
        gtile_load<NW>(gr, dy, g, b, ly, ox0, min(TW, g.LW - ox0), tid);
        patch_store<NT>(patch[0], pr, tid);
        gtile_to_lds<NW>(gbuf[0], gr, tid);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x, buf ^= 1) {
        int b, ly, ox0;
        tile_coords(g, tile, b, ly, ox0);
        const int nxt = tile + gridDim.x;
        if (nxt < g.ntiles) {
            int nb, nly, nox0;
            tile_coords(g, nxt, nb, nly, nox0);
            patch_load<NT>(pr, img, g, nb, nly, nox0, tid);
            gtile_load<NW>(gr, dy, g, nb, nly, nox0, min(TW, g.LW - nox0), tid);
        }
        const int npx = min(TW, g.LW - ox0);
        auto body = [&](auto FULL) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if (!decltype(FULL)::value && 32 * mb >= npx) break;
                const v16f h = conv_block(patch[buf], mb, lane, whi, wlo, bs);
                float d[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = 32 * mb + acc_pixel(r, hb);
                    const float tv = nsg_bf2f(gbuf[buf][j * gpitch(C) + c]);
                    const float x = SWAP ? tv : h[r], gv = SWAP ? h[r] : tv;
                    const float t = __builtin_fmaf(x, fs, off);
                    const float ge = t > 0.f ? gv : 0.f;
                    const float dv = __builtin_fmaf(sc, ge, -__builtin_fmaf(k1, x, k0));
                    if (SWAP) {
                        gbuf[buf][j * gpitch(C) + c] = nsg_f2bf(dv);      // over the element this lane just read; rows >= npx never leave LDS
                        d[r] = (decltype(FULL)::value || j < npx) ? fmaxf(t, 0.f) : 0.f;
                        cs += (decltype(FULL)::value || j < npx) ? dv : 0.f;     // column sum of du: the bias gradient of the conv in front
                    } else {
                        d[r] = (decltype(FULL)::value || j < npx) ? dv : 0.f;
                        cs += d[r];
                    }
                }
                // dw^T[tap][channel] += patch^T[tap][pixel] * d[pixel][channel]: two MFMAs of 16 pixels.  B: this lane's
                // registers 8s .. 8s+7 (K index 8hb + i <-> pixel acc_pixel(8s + i, hb)); A: the same pixels' values of this
                // lane's tap as bf16 (all lanes read -- lanes 16..31 a duplicate row -- and the padding rows are masked to zero).
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    v4u db, pa;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        db[i] = pack_bf16(d[8 * s + 2 * i], d[8 * s + 2 * i + 1]);
                        const int j0 = 32 * mb + acc_pixel(8 * s + 2 * i, hb), j1 = 32 * mb + acc_pixel(8 * s + 2 * i + 1, hb);
                        pa[i] = pack_bf16(patch[buf][tap_off + 2 * j0], patch[buf][tap_off + 2 * j1]) & tap_mask;
                    }
                    dwacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, db), dwacc, 0, 0, 0);
                }
            }
        };
        if (npx == TW) body(std::true_type{}); else body(std::false_type{});
        if (SWAP) {
            __syncthreads();                                              // the du tile is complete in gbuf[buf]
            gtile_store<NW>(dx_out, gbuf[buf], g, b, ly, ox0, npx, tid);
        }
        if (nxt < g.ntiles) {
            patch_store<NT>(patch[buf ^ 1], pr, tid);
            gtile_to_lds<NW>(gbuf[buf ^ 1], gr, tid);
        }
        __syncthreads();
    }
    // rows of dwacc are taps: register r < 8 of a lane in half hb is tap acc_pixel(r, hb) < 16
    float *dst = partial + ((size_t)blockIdx.x * C + c) * 17;
#pragma unroll
    for (int r = 0; r < 8; ++r) dst[acc_pixel(r, hb)] = dwacc[r];
    cs += __shfl_xor(cs, 32, 64);
    if (hb == 0) dst[16] = cs;
}

// ---------------------------------------------------------------------------------------------------------------
// The input layer by its TAP MOMENTS.  h[p][c] = b_c + sum_t w[c][t] x[p][t] is linear in the 16 patch values x[p][.] of an
// output pixel, so everything the BatchNorm around it needs that is LINEAR or QUADRATIC in h follows from
//     S[t] = sum_p x[p][t]   and   P[t][t'] = sum_p x[p][t] x[p][t']      (16 + 256 numbers, channel-independent, image only):
//   forward    mean_c = b_c + w_c . S / M,   var_c = w_c^T (P / M - S S^T / M^2) w_c          (no pass over h at all)
//   backward   dw[c][t] = sum_p x[p][t] dh[p][c] with dh = sc (g - dbeta / M - xhat dgamma / M) splits into
//              sc ( G[t][c] - dbeta_c / M S[t] - dgamma_c / M invstd_c ( (P w_c)[t] + b_c S[t] - mean_c S[t] ) ),
//              G[t][c] = sum_p x[p][t] g[p][c]  --  accumulated in the SAME pass over dy that forms dbeta = sum g and
//              dgamma = sum g xhat: the tensor is read once instead of twice.
// ---------------------------------------------------------------------------------------------------------------
// The moments are taken of the SHIFTED values x' = x - shift (every patch entry, padding included: an exact change of variables),
// shift = the mean of a fixed sample of the image, identical in every block: P' / M is then of the order of the covariance and
// the quadratic forms below lose nothing to a large image mean (var = w^T (P/M - S S^T / M^2) w cancels badly when mean^2 >> var).
static_assert(NSG_C1_MOMENTS == 16 * 17 + 1, "include/nsg.h");
constexpr int MOM_N = 16 * 17;          // mom[t * 17 + u] = P'[t][u] (u < 16), mom[t * 17 + 16] = S'[t];  mom[MOM_N] = shift
constexpr int MOM_SAMPLE = 256;

// the shift: mean of MOM_SAMPLE image values at a fixed stride, summed in a fixed tree -- the same number in every block / kernel
__device__ __forceinline__ float mom_shift(const float *__restrict__ img, int64_t nimg, float *red, int tid)
{
    const int64_t stride = nimg / MOM_SAMPLE > 0 ? nimg / MOM_SAMPLE : 1;
    float v = tid < MOM_SAMPLE ? img[((int64_t)tid * stride) % nimg] : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float sft = (((red[0] + red[1]) + red[2]) + red[3]) * (1.f / MOM_SAMPLE);
    __syncthreads();
    return sft;
}
constexpr int MOM_BLOCKS = 1024;

// P and S on the matrix cores: for the 16 pixels a wave takes from a tile, X^T X with X = [pixel][16 taps | 1] is ONE MFMA whose A
// and B operands are the same registers (lane = tap, 8 pixels per half-wave; lane 16 of B = the ones column that yields S),
// issued three times on hi / lo bf16 splits (hi*hi + hi*lo + lo*hi: x to 2^-17).  Block = 4 waves = a 64-pixel tile.
// partial[block][16][17]: row t = (P[t][0..15], S[t]).
constexpr int MOM_ROW = 17;
constexpr int MOM_GROUP = 8;         // tiles per iteration: their patch loads are in flight together (a single tile per iteration is
                                     // one global-memory latency per 64 pixels: 69 us for the 2.6 M pixels of the headline shape)
__global__ __launch_bounds__(256) void c1_tap_moments_kernel(const float *__restrict__ img, float *__restrict__ partial, const C1Geom g)
{
    constexpr int NT = 256;
    __shared__ __attribute__((aligned(16))) float patch[2][MOM_GROUP][4 * PP];
    __shared__ float red[4][16 * MOM_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hb = lane >> 5;
    const int tap_off = ((n >> 2) & 3) * PP + (n & 3);
    const unsigned tm = n < 16 ? 0xffffffffu : 0u;
    const v4u tap_mask = {tm, tm, tm, tm}, ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // (bf16 1.0 pairs)
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float sft = mom_shift(img, (int64_t)g.B * g.HH * g.WW, &red[0][0], tid);
    const int ngroups = (g.ntiles + MOM_GROUP - 1) / MOM_GROUP;
    float pr[MOM_GROUP][PatchRegs<NT>::N];
    auto load_group = [&](int grp) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < MOM_GROUP; ++q) {
            const int tile = min(grp * MOM_GROUP + q, g.ntiles - 1);      // (a tile past the end: a valid patch, masked below)
            int b, ly, ox0;
            tile_coords(g, tile, b, ly, ox0);
            patch_load<NT>(pr[q], img, g, b, ly, ox0, tid);
        }
    };
    auto store_group = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < MOM_GROUP; ++q) patch_store<NT>(patch[buf][q], pr[q], tid);
    };
    int buf = 0;
    if ((int)blockIdx.x < ngroups) { load_group(blockIdx.x); store_group(0); }
    __syncthreads();
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x, buf ^= 1) {
        const int nxt = grp + gridDim.x;
        if (nxt < ngroups) load_group(nxt);
#pragma unroll
        for (int q = 0; q < MOM_GROUP; ++q) {
            const int tile = grp * MOM_GROUP + q;
            int b, ly, ox0;
            tile_coords(g, min(tile, g.ntiles - 1), b, ly, ox0);
            const int npx = tile < g.ntiles ? min(TW, g.LW - ox0) : 0;
            // this lane's 8 pixels: j = 16 wave + 8 hb + i; its tap = n (lanes 16..31: zero rows; lane 16 of B: ones)
            bf16x8 ahi, alo, bhi, blo;
            if (npx == TW) {                    // full tile (the common case): masks on the packed operands only
                float x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = patch[buf][q][tap_off + 2 * (16 * wave + 8 * hb + i)] - sft;
                split8(x, ahi, alo);
                const v4u mh = __builtin_bit_cast(v4u, ahi) & tap_mask, ml = __builtin_bit_cast(v4u, alo) & tap_mask;
                ahi = __builtin_bit_cast(bf16x8, mh);
                alo = __builtin_bit_cast(bf16x8, ml);
                bhi = __builtin_bit_cast(bf16x8, n == 16 ? ones : mh);
                blo = __builtin_bit_cast(bf16x8, n == 16 ? v4u{0u, 0u, 0u, 0u} : ml);
            } else {
                // (masks as factors, not selects: a select lets hipcc sink each LDS read into its own branch, one LDS round trip at a time)
                float x[8], one[8];
                const float mt = n < 16 ? 1.f : 0.f, mo = n == 16 ? 1.f : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int j = 16 * wave + 8 * hb + i;
                    const float mp = j < npx ? 1.f : 0.f;
                    x[i] = (patch[buf][q][tap_off + 2 * j] - sft) * (mt * mp);
                    one[i] = mo * mp;
                }
                bf16x8 ohi, olo;
                split8(x, ahi, alo);
                split8(one, ohi, olo);             // (olo = 0)
                bhi = n == 16 ? ohi : ahi;
                blo = n == 16 ? olo : alo;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc, 0, 0, 0);
        }
        if (nxt < ngroups) store_group(buf ^ 1);
        __syncthreads();
    }
    // rows (taps) of acc: register r < 8 of a lane in half hb is row acc_pixel(r, hb) < 16; column = lane n (< 17 used)
    if (n < MOM_ROW) {
#pragma unroll
        for (int r = 0; r < 8; ++r) red[wave][acc_pixel(r, hb) * MOM_ROW + n] = acc[r];
    }
    __syncthreads();
    for (int k = tid; k < 16 * MOM_ROW; k += NT)
        partial[(size_t)blockIdx.x * 16 * MOM_ROW + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
}

// mom[k] = sum over blocks (block order, double)
__global__ __launch_bounds__(256) void c1_tap_moments_final_kernel(const float *__restrict__ partial, int nblocks, double *__restrict__ mom,
                                                                   const float *__restrict__ img, int64_t nimg)
{
    __shared__ double red[256];
    __shared__ float reds[4];
    const int tid = threadIdx.x;
    const float sft = mom_shift(img, nimg, reds, tid);
    if (blockIdx.x == 0 && tid == 0) mom[MOM_N] = (double)sft;
    const int k = blockIdx.x * 32 + (tid >> 3), j = tid & 7;          // 8 lanes share a moment
    double s = 0.0;
    if (k < MOM_N) {
        const int per = (nblocks + 7) / 8;
        const int b0 = j * per, b1 = min(nblocks, b0 + per);
        if (b1 > b0) s = nsg_strided_sum<double>(partial + (size_t)b0 * MOM_N + k, (size_t)MOM_N, b1 - b0);
    }
    red[tid] = s;
    __syncthreads();
    if (j != 0 || k >= MOM_N) return;
    s = 0.0;
    for (int q = 0; q < 8; ++q) s += red[tid + q];
    mom[k] = s;
}

__device__ __forceinline__ int mom_index(int t, int u) { return t * 17 + u; }      // P[t][u]
__device__ __forceinline__ int mom_sum(int t) { return t * 17 + 16; }              // S[t]

// one thread per channel: batch statistics of h from the tap moments (double), running statistics as nn.BatchNorm2d
__global__ __launch_bounds__(64) void c1_stats_from_moments_kernel(const double *__restrict__ mom, const float *__restrict__ w,
                                                                   const float *__restrict__ bias, int64_t M, int C, float eps, float momentum,
                                                                   float *__restrict__ mean, float *__restrict__ invstd,
                                                                   float *__restrict__ running_mean, float *__restrict__ running_var)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const double inv = 1.0 / (double)M;
    double wv[16], m[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { wv[t] = (double)w[(size_t)c * 16 + t]; m[t] = mom[mom_sum(t)] * inv; }     // m: means of the shifted taps
    const double sft = mom[MOM_N];
    double mu = bias ? (double)bias[c] : 0.0, var = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        mu += wv[t] * (m[t] + sft);
#pragma unroll
        for (int u = t; u < 16; ++u) {
            const double cov = mom[mom_index(t, u)] * inv - m[t] * m[u];
            var += (t == u ? 1.0 : 2.0) * wv[t] * wv[u] * cov;
        }
    }
    var = var > 0.0 ? var : 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    if (running_var) {
        const double var_u = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)var_u;
    }
}

// backward in ONE pass over dy: per block sums[2][C] = (sum g, invstd * sum g (h - mean)) and partial[c][17] = G^T (tap sums of
// x g; entry 16 unused), g = dy where the ReLU passed (exact in bf16).
//
// The vector ALU was this pass's bound (58 % busy, 3.1 TB/s: every wave split the same patch values into bf16 pairs, packed the
// same tap operands and carried two running sums per element), so everything that does not depend on the channel is now done
// ONCE per tile, by the thread that fetched it, on the way into LDS -- thread (pixel j, tap row ty) holds img[2ly-1+ty][2j-1 .. 2j+2]
// and writes, as (hi, lo) bf16 splits, the conv's A operand [pixel][tap] and its transpose [tap][pixel] for the weight-gradient
// MFMA -- and both sums come out of the matrix pipe: row 16 of the transposed operand is ones, so G[16][c] = sum g, and
//   sum_p g (h - mean) = sum_t w[c][t] G[t][c] + (bias - mean) G[16][c]      (h = bias + sum_t w_t x_t, all linear in x g)
// with G from BOTH halves of the split (x to ~2^-17: the weight gradient is now that exact, too).  Per element the ALU is left
// with the ReLU decision (the forward's fma) and a select on the raw bf16 bits.
template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 1 ? 3 : 4))) void c1m_bwd_onepass_kernel(
    const float *__restrict__ img, const float *__restrict__ w, const float *__restrict__ bias, const bf16_t *__restrict__ dy,
    const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ sums, float *__restrict__ partial, const C1Geom g)
{
    constexpr int NT = 64 * NW, C = 32 * NW;
    constexpr int AP = 16;              // bf16 per pixel row of the conv operand (its 16 taps)
    constexpr int QP = TW + 4;          // bf16 per tap row of the transposed operand (136 bytes: tap rows land 34 banks apart)
    constexpr int NJ = (4 * TW + NT - 1) / NT;      // (pixel, tap row) pairs per thread
    struct Operands {
        bf16_t ah[TW * AP], al[TW * AP];            // conv A operand: [pixel][tap], hi and lo halves of the split
        bf16_t ph[17 * QP], pl[16 * QP];            // its transpose [tap][pixel]; ph row 16 = 1.0
    };
    // ONE buffer each: the next tile waits in registers while this one is worked on, so a second LDS copy would only cost
    // residency -- and this pass is bound by the bytes its CU has in flight (four blocks per CU: 4.7 TB/s; three: 3.2)
    __shared__ __attribute__((aligned(16))) Operands ops;
    __shared__ __attribute__((aligned(16))) bf16_t gbuf[TW * gpitch(C)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hb = lane >> 5, c = 32 * wave + n;
    bf16x8 whi, wlo;
    load_w_operand(w, c, hb, whi, wlo);
    const float bs = bias ? bias[c] : 0.f;
    const float mu = mean[c], is = invstd[c];
    const float fs = is * gamma[c];
    const float off = __builtin_fmaf(-mu, fs, beta[c]);
    const int prow = n < 16 ? n : 16;                       // the lane's row of the transposed operand (rows above 16 are zero)
    const unsigned pmask = n <= 16 ? 0xffffffffu : 0u;
    v16f dwacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dwacc[r] = 0.f;
    for (int i = tid; i < TW; i += NT) ops.ph[16 * QP + i] = 0x3f80;        // the row of ones (never rewritten)

    // the thread's (pixel, tap row) pairs of a tile: four image values each
    auto taps_load = [&](float (&v)[NJ][4], int b, int ly, int ox0) {
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            const int e = tid + NT * q, j = e >> 2, ty = e & 3;
            const int y = 2 * ly - 1 + ty, x0 = 2 * (ox0 + j) - 1;
            const bool rowok = (e < 4 * TW) & (y >= 0) & (y < g.HH);
            const float *src = img + ((size_t)b * g.HH + (rowok ? y : 0)) * g.WW;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = x0 + k;
                const bool ok = rowok & (x >= 0) & (x < g.WW);
                const float t = src[ok ? x : 0];            // clamped, unconditional
                v[q][k] = ok ? t : 0.f;
            }
        }
    };
    auto taps_store = [&](Operands &o, const float (&v)[NJ][4]) {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            const int e = tid + NT * q, j = e >> 2, ty = e & 3;
            if (NJ * NT != 4 * TW && e >= 4 * TW) break;
            const unsigned h01 = pack_bf16(v[q][0], v[q][1]), h23 = pack_bf16(v[q][2], v[q][3]);
            const unsigned l01 = pack_bf16(v[q][0] - nsg_bitsf(h01 << 16), v[q][1] - nsg_bitsf(h01 & 0xffff0000u));
            const unsigned l23 = pack_bf16(v[q][2] - nsg_bitsf(h23 << 16), v[q][3] - nsg_bitsf(h23 & 0xffff0000u));
            *reinterpret_cast<v2u *>(o.ah + j * AP + 4 * ty) = v2u{h01, h23};
            *reinterpret_cast<v2u *>(o.al + j * AP + 4 * ty) = v2u{l01, l23};
            bf16_t *th = o.ph + (4 * ty) * QP + j, *tl = o.pl + (4 * ty) * QP + j;
            th[0] = (bf16_t)h01; th[QP] = (bf16_t)(h01 >> 16); th[2 * QP] = (bf16_t)h23; th[3 * QP] = (bf16_t)(h23 >> 16);
            tl[0] = (bf16_t)l01; tl[QP] = (bf16_t)(l01 >> 16); tl[2 * QP] = (bf16_t)l23; tl[3 * QP] = (bf16_t)(l23 >> 16);
        }
    };

    float tv[NJ][4];
    v4f gr[4];
    if ((int)blockIdx.x < g.ntiles) {
        int b, ly, ox0;
        tile_coords(g, blockIdx.x, b, ly, ox0);
        taps_load(tv, b, ly, ox0);
        gtile_load<NW>(gr, dy, g, b, ly, ox0, min(TW, g.LW - ox0), tid);
        taps_store(ops, tv);
        gtile_to_lds<NW>(gbuf, gr, tid);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        int b, ly, ox0;
        tile_coords(g, tile, b, ly, ox0);
        const int nxt = tile + gridDim.x;
        if (nxt < g.ntiles) {
            int nb, nly, nox0;
            tile_coords(g, nxt, nb, nly, nox0);
            taps_load(tv, nb, nly, nox0);
            gtile_load<NW>(gr, dy, g, nb, nly, nox0, min(TW, g.LW - nox0), tid);
        }
        const int npx = min(TW, g.LW - ox0);
        const Operands &o = ops;
        auto body = [&](auto FULL) {
#pragma unroll 1
            for (int mb = 0; mb < 2; ++mb) {
                if (!decltype(FULL)::value && 32 * mb >= npx) break;
                const bf16x8 ahi = *reinterpret_cast<const bf16x8 *>(o.ah + (32 * mb + n) * AP + 8 * hb);
                const bf16x8 alo = *reinterpret_cast<const bf16x8 *>(o.al + (32 * mb + n) * AP + 8 * hb);
                v16f h;
#pragma unroll
                for (int r = 0; r < 16; ++r) h[r] = 0.f;
                h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, whi, h, 0, 0, 0);       // conv_block's sequence: the forward's h, bit for bit
                h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, wlo, h, 0, 0, 0);
                h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, whi, h, 0, 0, 0);
                unsigned d[16];         // g as raw bf16 bits
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = 32 * mb + acc_pixel(r, hb);
                    const unsigned gv = gbuf[j * gpitch(C) + c];
                    bool pass = __builtin_fmaf(h[r] + bs, fs, off) > 0.f;
                    if (!decltype(FULL)::value) pass = pass & (j < npx);
                    d[r] = pass ? gv : 0u;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // k = 0 .. 7 of this half: pixels 16 s + 4 hb + {0 .. 3} and 16 s + 8 + 4 hb + {0 .. 3} of the block (acc_pixel)
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    const int px = 32 * mb + 16 * s + 4 * hb;
                    const v2u h0 = *reinterpret_cast<const v2u *>(o.ph + prow * QP + px), h1 = *reinterpret_cast<const v2u *>(o.ph + prow * QP + px + 8);
                    const v2u l0 = *reinterpret_cast<const v2u *>(o.pl + (prow & 15) * QP + px), l1 = *reinterpret_cast<const v2u *>(o.pl + (prow & 15) * QP + px + 8);
                    const unsigned lmask = n < 16 ? 0xffffffffu : 0u;
                    const v4u pah = {h0.x & pmask, h0.y & pmask, h1.x & pmask, h1.y & pmask};
                    const v4u pal = {l0.x & lmask, l0.y & lmask, l1.x & lmask, l1.y & lmask};
                    v4u db;
#pragma unroll
                    for (int i = 0; i < 4; ++i) db[i] = d[8 * s + 2 * i] | (d[8 * s + 2 * i + 1] << 16);
                    dwacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pal), __builtin_bit_cast(bf16x8, db), dwacc, 0, 0, 0);
                    dwacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pah), __builtin_bit_cast(bf16x8, db), dwacc, 0, 0, 0);
                }
            }
        };
        if (npx == TW) body(std::true_type{}); else body(std::false_type{});
        __syncthreads();                    // everyone has read this tile
        if (nxt < g.ntiles) {
            taps_store(ops, tv);
            gtile_to_lds<NW>(gbuf, gr, tid);
        }
        __syncthreads();
    }
    // rows of dwacc in this lane: taps acc_pixel(r, hb) for r < 8; row 16 (the ones) is r = 8 of half 0
    float *dst = partial + ((size_t)blockIdx.x * C + c) * 17;
    float wg = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int t = acc_pixel(r, hb);
        dst[t] = dwacc[r];
        wg = __builtin_fmaf(w[(size_t)c * 16 + t], dwacc[r], wg);
    }
    wg += __shfl_xor(wg, 32, 64);
    if (hb == 0) {
        const float s1 = dwacc[8];
        dst[16] = 0.f;
        float *sd = sums + (size_t)blockIdx.x * 2 * C;
        sd[c] = s1;
        sd[C + c] = __builtin_fmaf(bs - mu, s1, wg) * is;
    }
}

// dw[c][t] holds G[t][c] on entry; one thread per (c, t) finishes the weight gradient (double); dbias_c = sum_p dh[p][c]
__global__ __launch_bounds__(256) void c1m_onepass_fixup_kernel(const double *__restrict__ mom, const float *__restrict__ w,
                                                                const float *__restrict__ bias, const float *__restrict__ mean,
                                                                const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                                const float *__restrict__ dgamma, const float *__restrict__ dbeta, int64_t M,
                                                                int C, float *__restrict__ dw, float *__restrict__ dbias)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= C * 16) return;
    const int c = e >> 4, t = e & 15;
    const double inv = 1.0 / (double)M;
    const double is = (double)invstd[c], mu = (double)mean[c], b = bias ? (double)bias[c] : 0.0;
    const double sc = (double)gamma[c] * is;
    // with x = x' + shift: h = b' + sum_u w_u x'_u, b' = b + shift sum_u w_u, and sum_p x_t xhat = sum_p x'_t xhat (sum_p xhat = 0)
    const double sft = mom[MOM_N];
    double pw = 0.0, wsum = 0.0;         // (P' w_c)[t], sum_u w_u
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const double wu = (double)w[(size_t)c * 16 + u];
        pw += mom[mom_index(t, u)] * wu;
        wsum += wu;
    }
    const double bs = b + sft * wsum;
    const double Sp = mom[mom_sum(t)];                         // S'[t]
    const double St = Sp + (double)M * sft;                    // S[t]
    const double X = is * (pw + (bs - mu) * Sp);               // sum_p x[p][t] xhat[p][c]
    const double G = (double)dw[e];
    dw[e] = (float)(sc * (G - (double)dbeta[c] * inv * St - (double)dgamma[c] * inv * X));
    if (t == 0 && dbias) {               // sc (sum g - dbeta - dgamma / M sum xhat), sum xhat = invstd (sum h - M mean)
        double sh = (double)M * bs;
#pragma unroll
        for (int u = 0; u < 16; ++u) sh += (double)w[(size_t)c * 16 + u] * mom[mom_sum(u)];
        dbias[c] = (float)(-sc * (double)dgamma[c] * inv * is * (sh - (double)M * mu));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The output layer's forward: dots[m][t] = sum_c max(fma(u[m][c], fs[c], off[c]), 0) * w[c][t], m over all B*LH*LW pixels
// (the BatchNorm apply + ReLU happen on the operand's way into the MFMA: the activated tensor is never stored); the
// 16 tap products of a pixel are then scattered onto the image by col2im_c1_kernel (conv_api.hip) with bias and tanh.
// bf16 operands (a rounded to bf16 exactly as the backward rebuilds it, w rounded to bf16), fp32 accumulate.
// Block = 4 waves = 128 rows; MFMA out[tap][pixel] = w^T[tap][K = 16 channels] * a^T[K][pixel]: a lane is a pixel and
// reads its 8 consecutive channels of the k-step as one 16-byte LDS read of the row-major tile.
// ---------------------------------------------------------------------------------------------------------------
template <int NW>     // C = 32 NW
__global__ __launch_bounds__(256) void bnrelu_dots_kernel(const bf16_t *__restrict__ u, const float *__restrict__ mean,
                                                          const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, const float *__restrict__ w,
                                                          float *__restrict__ dots, int64_t M)
{
    constexpr int C = 32 * NW, KS = C / 16, ROWS = 128, CPR = C / 8, PIECES = ROWS * CPR / 256;   // 16-byte pieces per thread
    __shared__ __attribute__((aligned(16))) bf16_t tile[ROWS * gpitch(C)];
    __shared__ __attribute__((aligned(16))) float cfs[C], coff[C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hb = lane >> 5;
    for (int c = tid; c < C; c += 256) {
        const float fs = invstd[c] * gamma[c];
        cfs[c] = fs;
        coff[c] = __builtin_fmaf(-mean[c], fs, beta[c]);
    }
    // A operand per k-step: lane (tap n, hb) holds w[16 ks + 8 hb + i][n], i = 0..7, as bf16 (rows 16..31 zero)
    v4u wa[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c0 = 16 * ks + 8 * hb + 2 * i;       // (unconditional loads, then the select: a load under a lane condition
            const unsigned pk = pack_bf16(w[(size_t)c0 * 16 + (n & 15)], w[(size_t)(c0 + 1) * 16 + (n & 15)]);   // gets its own branch
            wa[ks][i] = n < 16 ? pk : 0u;                  // and s_waitcnt vmcnt(0): 4 KS memory round trips in a row)
        }
    const int64_t ntiles = (M + ROWS - 1) / ROWS;
    v4f pr[PIECES];
    auto prefetch = [&](int64_t t) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int q = tid + 256 * i;
            const int row = q / CPR, cc = q - row * CPR;
            const int64_t m = t * ROWS + row;
            pr[i] = *reinterpret_cast<const v4f *>(u + (size_t)(m < M ? m : M - 1) * C + 8 * cc);   // clamped
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int q = tid + 256 * i;
            const int row = q / CPR, cc = q - row * CPR;
            *reinterpret_cast<v4f *>(tile + row * gpitch(C) + 8 * cc) = pr[i];
        }
    };
    if ((int64_t)blockIdx.x < ntiles) { prefetch(blockIdx.x); stage(); }
    __syncthreads();
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t nxt = t + gridDim.x;
        if (nxt < ntiles) prefetch(nxt);
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bf16_t *rowp = tile + (32 * wave + n) * gpitch(C) + 8 * hb;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const v4f raw = *reinterpret_cast<const v4f *>(rowp + 16 * ks);
            const v4f f0 = *reinterpret_cast<const v4f *>(cfs + 16 * ks + 8 * hb), f1 = *reinterpret_cast<const v4f *>(cfs + 16 * ks + 8 * hb + 4);
            const v4f o0 = *reinterpret_cast<const v4f *>(coff + 16 * ks + 8 * hb), o1 = *reinterpret_cast<const v4f *>(coff + 16 * ks + 8 * hb + 4);
            const float fs8[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
            const float of8[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
            float uv[8];
            Elem<bf16_t>::unpack16(raw, uv);
            v4u a;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = pack_bf16(fmaxf(__builtin_fmaf(uv[2 * i], fs8[2 * i], of8[2 * i]), 0.f),
                                 fmaxf(__builtin_fmaf(uv[2 * i + 1], fs8[2 * i + 1], of8[2 * i + 1]), 0.f));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[ks]), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
        }
        // lane (pixel n, hb): registers r < 8 are taps (r & 3) + 8 (r >> 2) + 4 hb: two runs of 4 consecutive taps
        const int64_t m = t * ROWS + 32 * wave + n;
        if (m < M) {
            *reinterpret_cast<v4f *>(dots + (size_t)m * 16 + 4 * hb) = v4f{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<v4f *>(dots + (size_t)m * 16 + 8 + 4 * hb) = v4f{acc[4], acc[5], acc[6], acc[7]};
        }
        __syncthreads();                 // everyone is done reading the tile
        if (nxt < ntiles) stage();
        __syncthreads();
    }
}

}  // namespace

// ---- launchers (stencil_c1.hip's C ABI picks these for bf16 tensors with C = 32, 64, 96 or 128) ----
bool nsg_c1m_supported(int C) { return (C % 32 == 0 && C >= 32 && C <= 128) || C == 256; }

#define NSG_C1M_DISPATCH(KERNEL, ...)                                                                                   \
    switch (C / 32) {                                                                                                   \
    case 1: hipLaunchKernelGGL((KERNEL<1>), dim3(blocks), dim3(64), 0, s, __VA_ARGS__); break;                          \
    case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(blocks), dim3(128), 0, s, __VA_ARGS__); break;                         \
    case 3: hipLaunchKernelGGL((KERNEL<3>), dim3(blocks), dim3(192), 0, s, __VA_ARGS__); break;                         \
    case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(blocks), dim3(512), 0, s, __VA_ARGS__); break;                         \
    default: hipLaunchKernelGGL((KERNEL<4>), dim3(blocks), dim3(256), 0, s, __VA_ARGS__); break;                        \
    }

#define NSG_C1M_DISPATCH2(KERNEL, FLAG, ...)                                                                            \
    switch (C / 32) {                                                                                                   \
    case 1: hipLaunchKernelGGL((KERNEL<1, FLAG>), dim3(blocks), dim3(64), 0, s, __VA_ARGS__); break;                    \
    case 2: hipLaunchKernelGGL((KERNEL<2, FLAG>), dim3(blocks), dim3(128), 0, s, __VA_ARGS__); break;                   \
    case 3: hipLaunchKernelGGL((KERNEL<3, FLAG>), dim3(blocks), dim3(192), 0, s, __VA_ARGS__); break;                   \
    case 8: hipLaunchKernelGGL((KERNEL<8, FLAG>), dim3(blocks), dim3(512), 0, s, __VA_ARGS__); break;                   \
    default: hipLaunchKernelGGL((KERNEL<4, FLAG>), dim3(blocks), dim3(256), 0, s, __VA_ARGS__); break;                  \
    }

int nsg_launch_c1m_stats(const float *img, const float *w, const float *bias, float *tiles, int blocks, int B, int LH, int LW, int HH, int WW,
                         int C, hipStream_t s)
{
    const C1Geom g = make_geom(B, LH, LW, HH, WW, C);
    NSG_C1M_DISPATCH(c1m_stats_kernel, img, w, bias, tiles, g)
    return nsg_check_launch("c1m_stats_kernel");
}

int nsg_launch_c1m_apply(const float *img, const float *w, const float *bias, const float *mean, const float *invstd, const float *gamma,
                         const float *beta, void *out, int blocks, int B, int LH, int LW, int HH, int WW, int C, hipStream_t s)
{
    const C1Geom g = make_geom(B, LH, LW, HH, WW, C);
    NSG_C1M_DISPATCH(c1m_apply_kernel, img, w, bias, mean, invstd, gamma, beta, reinterpret_cast<bf16_t *>(out), g)
    return nsg_check_launch("c1m_apply_kernel");
}

int nsg_launch_c1m_bwd_sums(const float *img, const float *w, const float *bias, const void *dy, const float *mean, const float *invstd,
                            const float *gamma, const float *beta, float *partial, int blocks, int B, int LH, int LW, int HH, int WW, int C,
                            hipStream_t s)
{
    const C1Geom g = make_geom(B, LH, LW, HH, WW, C);
    NSG_C1M_DISPATCH2(c1m_bwd_sums_kernel, false, img, w, bias, reinterpret_cast<const bf16_t *>(dy), mean, invstd, gamma, beta, partial, g)
    return nsg_check_launch("c1m_bwd_sums_kernel");
}

int nsg_launch_c1m_bwd_wgrad(const float *img, const float *w, const float *bias, const void *dy, const float *mean, const float *invstd,
                             const float *gamma, const float *beta, const float *dgamma, const float *dbeta, float inv_m, float *partial,
                             int blocks, int B, int LH, int LW, int HH, int WW, int C, hipStream_t s)
{
    const C1Geom g = make_geom(B, LH, LW, HH, WW, C);
    NSG_C1M_DISPATCH2(c1m_bwd_wgrad_kernel, false, img, w, bias, reinterpret_cast<const bf16_t *>(dy), mean, invstd, gamma, beta, dgamma, dbeta,
                      inv_m, partial, (bf16_t *)nullptr, g)
    return nsg_check_launch("c1m_bwd_wgrad_kernel");
}

// ---- the input layer by its tap moments ----
size_t nsg_c1m_moments_bytes() { return nsg_align_up((size_t)MOM_BLOCKS * MOM_N * sizeof(float), 256) + nsg_align_up((size_t)NSG_C1_MOMENTS * sizeof(double), 256); }

// ws: nsg_c1m_moments_bytes() bytes; the moments (NSG_C1_MOMENTS doubles) go to mom_dst, or into ws when that is null; *mom_out = where
int nsg_launch_c1m_moments(const float *img, int B, int LH, int LW, int HH, int WW, void *ws, double *mom_dst, const double **mom_out, hipStream_t s)
{
    const C1Geom g = make_geom(B, LH, LW, HH, WW, 32);
    float *partial = reinterpret_cast<float *>(ws);
    double *mom = mom_dst ? mom_dst : reinterpret_cast<double *>(reinterpret_cast<char *>(ws) + nsg_align_up((size_t)MOM_BLOCKS * MOM_N * sizeof(float), 256));
    int64_t blocks = (g.ntiles + MOM_GROUP - 1) / MOM_GROUP;
    if (blocks > MOM_BLOCKS) blocks = MOM_BLOCKS;
    hipLaunchKernelGGL(c1_tap_moments_kernel, dim3((unsigned)blocks), dim3(256), 0, s, img, partial, g);
    hipLaunchKernelGGL(c1_tap_moments_final_kernel, dim3((MOM_N + 31) / 32), dim3(256), 0, s, partial, (int)blocks, mom, img,
                       (int64_t)B * HH * WW);
    *mom_out = mom;
    return nsg_check_launch("c1_tap_moments_kernel");
}

int nsg_launch_c1m_stats_from_moments(const double *mom, const float *w, const float *bias, int64_t M, int C, float eps, float momentum,
                                      float *mean, float *invstd, float *running_mean, float *running_var, hipStream_t s)
{
    hipLaunchKernelGGL(c1_stats_from_moments_kernel, dim3((C + 63) / 64), dim3(64), 0, s, mom, w, bias, M, C, eps, momentum, mean, invstd,
                       running_mean, running_var);
    return nsg_check_launch("c1_stats_from_moments_kernel");
}

int nsg_launch_c1m_bwd_onepass(const float *img, const float *w, const float *bias, const void *dy, const float *mean, const float *invstd,
                               const float *gamma, const float *beta, float *sums, float *partial17, int blocks, int B, int LH, int LW,
                               int HH, int WW, int C, hipStream_t s)
{
    const C1Geom g = make_geom(B, LH, LW, HH, WW, C);
    NSG_C1M_DISPATCH(c1m_bwd_onepass_kernel, img, w, bias, reinterpret_cast<const bf16_t *>(dy), mean, invstd, gamma, beta, sums, partial17, g)
    return nsg_check_launch("c1m_bwd_onepass_kernel");
}

int nsg_launch_c1m_onepass_fixup(const double *mom, const float *w, const float *bias, const float *mean, const float *invstd,
                                 const float *gamma, const float *dgamma, const float *dbeta, int64_t M, int C, float *dw, float *dbias,
                                 hipStream_t s)
{
    hipLaunchKernelGGL(c1m_onepass_fixup_kernel, dim3((C * 16 + 255) / 256), dim3(256), 0, s, mom, w, bias, mean, invstd, gamma, dgamma, dbeta,
                       M, C, dw, dbias);
    return nsg_check_launch("c1m_onepass_fixup_kernel");
}

// ---- the output layer BatchNorm -> ReLU -> ConvTranspose2d(C, 1, 4, 2, 1) (conv_api.hip: nsg_bn_relu_c1convt_*) ----
int nsg_launch_bnrelu_dots(const void *u, const float *mean, const float *invstd, const float *gamma, const float *beta, const float *w,
                           float *dots, int64_t M, int C, hipStream_t s)
{
    int64_t blocks = (M + 127) / 128;
    const int64_t cap = C > 128 ? 512 : 1024;      // the blocks resident at once (LDS): every block pays the weight prologue once
    if (blocks > cap) blocks = cap;
    if (blocks < 1) return NSG_OK;
    const bf16_t *ub = reinterpret_cast<const bf16_t *>(u);
    switch (C / 32) {
    case 1: hipLaunchKernelGGL((bnrelu_dots_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, ub, mean, invstd, gamma, beta, w, dots, M); break;
    case 2: hipLaunchKernelGGL((bnrelu_dots_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, ub, mean, invstd, gamma, beta, w, dots, M); break;
    case 3: hipLaunchKernelGGL((bnrelu_dots_kernel<3>), dim3((unsigned)blocks), dim3(256), 0, s, ub, mean, invstd, gamma, beta, w, dots, M); break;
    case 8: hipLaunchKernelGGL((bnrelu_dots_kernel<8>), dim3((unsigned)blocks), dim3(256), 0, s, ub, mean, invstd, gamma, beta, w, dots, M); break;
    default: hipLaunchKernelGGL((bnrelu_dots_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, ub, mean, invstd, gamma, beta, w, dots, M); break;
    }
    return nsg_check_launch("bnrelu_dots_kernel");
}

// dimg: the gradient image [B][2 LH][2 LW] fp32 (w.r.t. the transposed conv's output); u, du: [B][LH][LW][C] bf16
int nsg_launch_c1m_out_bwd_sums(const float *dimg, const float *w, const void *u, const float *mean, const float *invstd, const float *gamma,
                                const float *beta, float *partial, int blocks, int B, int LH, int LW, int C, hipStream_t s)
{
    const C1Geom g = make_geom(B, LH, LW, 2 * LH, 2 * LW, C);
    NSG_C1M_DISPATCH2(c1m_bwd_sums_kernel, true, dimg, w, (const float *)nullptr, reinterpret_cast<const bf16_t *>(u), mean, invstd, gamma, beta,
                      partial, g)
    return nsg_check_launch("c1m_bwd_sums_kernel<out>");
}

int nsg_launch_c1m_out_bwd_apply(const float *dimg, const float *w, const void *u, const float *mean, const float *invstd, const float *gamma,
                                 const float *beta, const float *dgamma, const float *dbeta, float inv_m, float *partial17, void *du, int blocks,
                                 int B, int LH, int LW, int C, hipStream_t s)
{
    const C1Geom g = make_geom(B, LH, LW, 2 * LH, 2 * LW, C);
    NSG_C1M_DISPATCH2(c1m_bwd_wgrad_kernel, true, dimg, w, (const float *)nullptr, reinterpret_cast<const bf16_t *>(u), mean, invstd, gamma, beta,
                      dgamma, dbeta, inv_m, partial17, reinterpret_cast<bf16_t *>(du), g)
    return nsg_check_launch("c1m_bwd_wgrad_kernel<out>");
}
