// The two single-channel layers of the model -- encoder.0 = Conv2d(1, D, 4, 2, 1) (src/models.py:165) and
// decoder.6 = ConvTranspose2d(D, 1, 4, 2, 1) (src/models.py:182) -- as HBM-bound stencil kernels.
//
// Both relate a one-channel fp32 image [B][HH][WW] to a D-channel tensor t [B][LH][LW][C] on the half-resolution
// grid (LH, LW) = (HH/2, WW/2) through the same 16 taps: patch(ly, lx)[kh*4+kw] = img[2ly-1+kh][2lx-1+kw].
//   c1_stencil_fwd:   t[pix][c]  = bias[c] + sum_t patch(pix)[t] * w[c][t]     (encoder.0 forward; decoder.6 data gradient)
//   c1_stencil_wgrad: dw[c][t]   = sum_pix t[pix][c] * patch(pix)[t]           (encoder.0 / decoder.6 weight gradient)
//                     colsum[c]  = sum_pix t[pix][c]                            (encoder.0 bias gradient)
// 16 multiply-adds per tensor element is far below the machine balance, so the kernels are organised around the one
// big stream (t: 16-byte or 8-byte accesses, a half-wave covers one pixel's channels contiguously); the image patch
// rows of a tile sit in LDS and are read as wave-broadcasts; the 64 weights (or 64+4 accumulators) a thread needs
// stay in registers across a persistent loop over tiles.  w is the parameter's own layout: Conv2d (C,1,4,4) and
// ConvTranspose2d (C,1,4,4) are both [c][16].
#include "nsg_common.h"
#include "c1_geom.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

// The 4 image rows x (2*TW+2) columns a tile needs (out-of-image positions are zero: the conv padding), in two halves so
// that the NEXT tile's global loads are in flight while the current tile is computed: patch_load -> registers,
// patch_store -> LDS.  4*(2*TW+2) = 520 values = 3 per thread of 256 (the last partly).
constexpr int PATCH_PER_THREAD = (PATCH_VALS + 255) / 256;
__device__ __forceinline__ void patch_load(float (&r)[PATCH_PER_THREAD], const float *__restrict__ img, const C1Geom &g, int b, int ly, int ox0, int tid)
{
    constexpr int NCOL = 2 * TW + 2;
#pragma unroll
    for (int q = 0; q < PATCH_PER_THREAD; ++q) {
        const int i = tid + 256 * q;
        const int rr = i / NCOL, cix = i - rr * NCOL;
        const int y = 2 * ly - 1 + rr, x = 2 * ox0 - 1 + cix;
        const bool ok = (i < PATCH_VALS) & (y >= 0) & (y < g.HH) & (x >= 0) & (x < g.WW);
        const float v = img[ok ? ((size_t)b * g.HH + y) * g.WW + x : 0];     // clamped, unconditional
        r[q] = ok ? v : 0.f;
    }
}
__device__ __forceinline__ void patch_store(float *patch, const float (&r)[PATCH_PER_THREAD], int tid)
{
    constexpr int NCOL = 2 * TW + 2;
#pragma unroll
    for (int q = 0; q < PATCH_PER_THREAD; ++q) {
        const int i = tid + 256 * q;
        if (i < PATCH_VALS) {
            const int rr = i / NCOL, cix = i - rr * NCOL;
            patch[rr * PP + cix] = r[q];
        }
    }
}

// pixel j of the tile: its 16 taps as 8 pairs (kw pairs), tap index t = 4*kh + kw
__device__ __forceinline__ void read_taps(const float *patch, int j, v2f tp[8])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        tp[2 * r] = *reinterpret_cast<const v2f *>(patch + r * PP + 2 * j);
        tp[2 * r + 1] = *reinterpret_cast<const v2f *>(patch + r * PP + 2 * j + 2);
    }
}


// The 16 weights of a thread's 4 channels as channel pairs: wp[t][0] = (w[c0][t], w[c0+1][t]), wp[t][1] = (w[c0+2][t], w[c0+3][t])
__device__ __forceinline__ void load_w4(const float *__restrict__ w, int c0, bool active, v2f (&wp)[16][2])
{
    if (active) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4f a0 = *reinterpret_cast<const v4f *>(w + (size_t)(c0 + 0) * 16 + 4 * q);
            const v4f a1 = *reinterpret_cast<const v4f *>(w + (size_t)(c0 + 1) * 16 + 4 * q);
            const v4f a2 = *reinterpret_cast<const v4f *>(w + (size_t)(c0 + 2) * 16 + 4 * q);
            const v4f a3 = *reinterpret_cast<const v4f *>(w + (size_t)(c0 + 3) * 16 + 4 * q);
            wp[4 * q + 0][0] = v2f{a0.x, a1.x}; wp[4 * q + 0][1] = v2f{a2.x, a3.x};
            wp[4 * q + 1][0] = v2f{a0.y, a1.y}; wp[4 * q + 1][1] = v2f{a2.y, a3.y};
            wp[4 * q + 2][0] = v2f{a0.z, a1.z}; wp[4 * q + 2][1] = v2f{a2.z, a3.z};
            wp[4 * q + 3][0] = v2f{a0.w, a1.w}; wp[4 * q + 3][1] = v2f{a2.w, a3.w};
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) { wp[t][0] = v2f{0.f, 0.f}; wp[t][1] = v2f{0.f, 0.f}; }
    }
}

// One pixel's 4 channel values from its taps: tap-ordered multiply-add chain from zero, bias added last.  EVERY kernel of
// this file that needs the convolution's value goes through here, so a recomputed value is the stored one bit for bit.
__device__ __forceinline__ void conv_taps(const v2f (&tp)[8], const v2f (&wp)[16][2], const v2f (&bv)[2], v2f &a0, v2f &a1)
{
    a0 = v2f{0.f, 0.f};
    a1 = v2f{0.f, 0.f};
#pragma unroll
    for (int t2 = 0; t2 < 8; ++t2) {
        const v2f lo = {tp[t2].x, tp[t2].x}, hi = {tp[t2].y, tp[t2].y};
        a0 = __builtin_elementwise_fma(lo, wp[2 * t2][0], a0);
        a1 = __builtin_elementwise_fma(lo, wp[2 * t2][1], a1);
        a0 = __builtin_elementwise_fma(hi, wp[2 * t2 + 1][0], a0);
        a1 = __builtin_elementwise_fma(hi, wp[2 * t2 + 1][1], a1);
    }
    a0 += bv[0];
    a1 += bv[1];
}

template <typename TO> __device__ __forceinline__ void store4(TO *dst, float x0, float x1, float x2, float x3)
{
    if constexpr (sizeof(TO) == 4) {
        *reinterpret_cast<v4f *>(dst) = v4f{x0, x1, x2, x3};
    } else {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const v2u pk = {nsg_pack_bf16(x0, x1), nsg_pack_bf16(x2, x3)};
        *reinterpret_cast<v2u *>(dst) = pk;
    }
}
template <typename TI> __device__ __forceinline__ void load4(const TI *src, float (&o)[4])
{
    if constexpr (sizeof(TI) == 4) {
        const v4f v = *reinterpret_cast<const v4f *>(src);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const v2u pk = *reinterpret_cast<const v2u *>(src);
        o[0] = nsg_bitsf(pk.x << 16); o[1] = nsg_bitsf(pk.x & 0xffff0000u);
        o[2] = nsg_bitsf(pk.y << 16); o[3] = nsg_bitsf(pk.y & 0xffff0000u);
    }
}

template <typename TO>
__global__ __launch_bounds__(256) void c1_stencil_fwd_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                             const float *__restrict__ bias, TO *__restrict__ out, const C1Geom g)
{
    __shared__ __attribute__((aligned(16))) float patch[2][4 * PP];
    const int tid = threadIdx.x;
    const int G = g.C >> 2;             // channel groups of 4
    const int SL = 256 / G;             // pixel slots
    const int cg = tid % G, sl = tid / G;
    const bool active = sl < SL;
    const int c0 = cg * 4;

    v2f wp[16][2];
    load_w4(w, c0, active, wp);
    v2f bv[2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (active && bias) { bv[0] = v2f{bias[c0], bias[c0 + 1]}; bv[1] = v2f{bias[c0 + 2], bias[c0 + 3]}; }

    int buf = 0;
    float pr[PATCH_PER_THREAD];
    if ((int)blockIdx.x < g.ntiles) {
        int b, ly, ox0;
        tile_coords(g, blockIdx.x, b, ly, ox0);
        patch_load(pr, img, g, b, ly, ox0, tid);
        patch_store(patch[0], pr, tid);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x, buf ^= 1) {
        int b, ly, ox0;
        tile_coords(g, tile, b, ly, ox0);
        const int nxt = tile + gridDim.x;
        if (nxt < g.ntiles) {                 // the next tile's image rows travel while this tile is computed
            int nb, nly, nox0;
            tile_coords(g, nxt, nb, nly, nox0);
            patch_load(pr, img, g, nb, nly, nox0, tid);
        }
        const int npx = min(TW, g.LW - ox0);
        TO *orow = out + (((size_t)b * g.LH + ly) * g.LW + ox0) * g.C + c0;
        for (int j = sl; active && j < npx; j += SL) {
            v2f tp[8];
            read_taps(patch[buf], j, tp);
            v2f a0, a1;
            conv_taps(tp, wp, bv, a0, a1);
            store4<TO>(orow + (size_t)j * g.C, a0.x, a0.y, a1.x, a1.y);
        }
        if (nxt < g.ntiles) patch_store(patch[buf ^ 1], pr, tid);   // (its last readers passed the barrier below one tile ago)
        __syncthreads();
    }
}

// partial[block][c][17]: 16 tap sums + the column sum of t, over the block's tiles, reduced over the block's pixel
// slots in slot order (fixed order: bitwise reproducible).  The stream of t is the whole cost: four pixels' loads
// are in flight per thread before their multiply-adds start.
template <typename TI, bool RELU>
__global__ __launch_bounds__(256) void c1_stencil_wgrad_kernel(const float *__restrict__ img, const TI *__restrict__ t,
                                                               float *__restrict__ partial, const C1Geom g)
{
    __shared__ __attribute__((aligned(16))) float patch[2][4 * PP];
    __shared__ float red[17 * 256];
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
    const int G = g.C >> 2;
    const int SL = 256 / G;
    const int cg = tid % G, sl = tid / G;
    const bool active = sl < SL;
    const int c0 = cg * 4;

    v2f acc[4][8];      // [channel][tap pair]
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[c][k] = v2f{0.f, 0.f};

    int buf = 0;
    float pr[PATCH_PER_THREAD];
    if ((int)blockIdx.x < g.ntiles) {
        int b, ly, ox0;
        tile_coords(g, blockIdx.x, b, ly, ox0);
        patch_load(pr, img, g, b, ly, ox0, tid);
        patch_store(patch[0], pr, tid);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x, buf ^= 1) {
        int b, ly, ox0;
        tile_coords(g, tile, b, ly, ox0);
        const int nxt = tile + gridDim.x;
        if (nxt < g.ntiles) {
            int nb, nly, nox0;
            tile_coords(g, nxt, nb, nly, nox0);
            patch_load(pr, img, g, nb, nly, nox0, tid);
        }
        const int npx = min(TW, g.LW - ox0);
        const TI *trow = t + (((size_t)b * g.LH + ly) * g.LW + ox0) * g.C + c0;
        for (int j0 = sl; active && j0 < npx; j0 += 4 * SL) {
            float tv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * SL;
                const int jc = j < npx ? j : j0;          // clamped: always a valid address
                if constexpr (sizeof(TI) == 4) {
                    const v4f v = *reinterpret_cast<const v4f *>(trow + (size_t)jc * g.C);
                    tv[u][0] = v.x; tv[u][1] = v.y; tv[u][2] = v.z; tv[u][3] = v.w;
                } else {
                    const v2u pk = *reinterpret_cast<const v2u *>(trow + (size_t)jc * g.C);
                    const unsigned u0 = pk.x, u1 = pk.y;
                    tv[u][0] = nsg_bitsf(u0 << 16); tv[u][1] = nsg_bitsf(u0 & 0xffff0000u);
                    tv[u][2] = nsg_bitsf(u1 << 16); tv[u][3] = nsg_bitsf(u1 & 0xffff0000u);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * SL;
                if (j < npx) {
                    v2f tp[8];
                    read_taps(patch[buf], j, tp);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float x = RELU ? fmaxf(tv[u][c], 0.f) : tv[u][c];
                        const v2f tc = {x, x};
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[c][k] = __builtin_elementwise_fma(tc, tp[k], acc[c][k]);
                        cs[c] += x;
                    }
                }
            }
        }
        if (nxt < g.ntiles) patch_store(patch[buf ^ 1], pr, tid);
        __syncthreads();
    }
    // ---- reduce over the block's slots, one of the thread's 4 channels per pass: red[v][tid], v = tap (16 = column sum) ----
    float *dst = partial + (size_t)blockIdx.x * g.C * 17;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[(2 * k) * 256 + tid] = active ? acc[c][k].x : 0.f;
            red[(2 * k + 1) * 256 + tid] = active ? acc[c][k].y : 0.f;
        }
        red[16 * 256 + tid] = active ? cs[c] : 0.f;
        __syncthreads();
        for (int e = tid; e < G * 17; e += 256) {      // e = v * G + cg'
            const int v = e / G, cgp = e - v * G;
            float s = 0.f;
            for (int q = 0; q < SL; ++q) s += red[v * 256 + q * G + cgp];
            dst[(size_t)(cgp * 4 + c) * 17 + v] = s;
        }
    }
}

// dw[c][t] = sum over blocks (fixed order, double), optional colsum[c]
__global__ __launch_bounds__(256) void c1_stencil_wgrad_final_kernel(const float *__restrict__ partial, int nblocks, int C,
                                                                     float *__restrict__ dw, float *__restrict__ colsum)
{
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int j = tid & 7;                       // 8 lanes share an output
    const int e = blockIdx.x * 32 + (tid >> 3);  // output index in [0, C*17)
    const int total = C * 17;
    double s = 0.0;
    if (e < total) {
        const int per = (nblocks + 7) / 8;
        const int b0 = j * per, b1 = min(nblocks, b0 + per);
        if (b1 > b0) s = nsg_strided_sum<double>(partial + (size_t)b0 * total + e, (size_t)total, b1 - b0);
    }
    red[tid] = s;
    __syncthreads();
    if (j != 0 || e >= total) return;
    s = 0.0;
    for (int k = 0; k < 8; ++k) s += red[tid + k];
    const int c = e / 17, tap = e - c * 17;
    if (tap < 16) dw[c * 16 + tap] = (float)s;
    else if (colsum) colsum[c] = (float)s;
}


// ------------------------------------------------------------------------------------------------------------------
// encoder.0 .. encoder.2 as ONE layer: Conv2d(1, C, 4, 2, 1) -> BatchNorm2d(C) -> ReLU (src/models.py:165-167).
// The convolution's output h costs 16 multiply-adds per element from an image 2C times smaller than itself, so it is
// never stored: each pass that needs h recomputes it (conv_taps) from the image patch in LDS.
//   forward   stats pass  (image -> per-block (count, sum, M2) of h)              no tensor traffic at all
//             apply pass  (image -> y = relu((h - mean) * invstd*gamma + beta))   one tensor write
//   backward  sums pass   (image, dy -> per-block sum g, sum g*xhat; g = dy masked by the ReLU)        one tensor read
//             grad pass   (image, dy -> dh = BatchNorm backward of g, folded straight into the weight-gradient
//                          accumulators dw[c][t] += dh * patch[t] and the bias gradient)                one tensor read
// against three tensor writes and eight tensor reads of the unfused sequence.
// ------------------------------------------------------------------------------------------------------------------

// The persistent tile loop shared by the passes: body(b, ly, ox0, npx, patch of this tile).
template <typename F>
__device__ __forceinline__ void tile_loop(const float *__restrict__ img, const C1Geom &g, float (*patch)[4 * PP], int tid, F body)
{
    int buf = 0;
    float pr[PATCH_PER_THREAD];
    if ((int)blockIdx.x < g.ntiles) {
        int b, ly, ox0;
        tile_coords(g, blockIdx.x, b, ly, ox0);
        patch_load(pr, img, g, b, ly, ox0, tid);
        patch_store(patch[0], pr, tid);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x, buf ^= 1) {
        int b, ly, ox0;
        tile_coords(g, tile, b, ly, ox0);
        const int nxt = tile + gridDim.x;
        if (nxt < g.ntiles) {
            int nb, nly, nox0;
            tile_coords(g, nxt, nb, nly, nox0);
            patch_load(pr, img, g, nb, nly, nox0, tid);
        }
        body(b, ly, ox0, min(TW, g.LW - ox0), patch[buf]);
        if (nxt < g.ntiles) patch_store(patch[buf ^ 1], pr, tid);
        __syncthreads();
    }
}

// tiles[block][3][C] = (count, sum, M2 about the block mean) of h over the block's pixels (bn_stats_tiles_final_kernel pools
// them in double).  Per thread one pass about a pivot (its first value); the block's pixel slots are pooled in slot order.
__global__ __launch_bounds__(256) void c1_bn_stats_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                          const float *__restrict__ bias, float *__restrict__ tiles, const C1Geom g)
{
    __shared__ __attribute__((aligned(16))) float patch[2][4 * PP];
    __shared__ float red[9 * 256];
    const int tid = threadIdx.x;
    const int G = g.C >> 2, SL = 256 / G;
    const int cg = tid % G, sl = tid / G;
    const bool active = sl < SL;
    const int c0 = cg * 4;
    v2f wp[16][2];
    load_w4(w, c0, active, wp);
    v2f bv[2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (active && bias) { bv[0] = v2f{bias[c0], bias[c0 + 1]}; bv[1] = v2f{bias[c0 + 2], bias[c0 + 3]}; }

    float pv[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    int n = 0;
    tile_loop(img, g, patch, tid, [&](int, int, int, int npx, const float *pt) {
        for (int j = sl; active && j < npx; j += SL) {
            v2f tp[8], a0, a1;
            read_taps(pt, j, tp);
            conv_taps(tp, wp, bv, a0, a1);
            const float h[4] = {a0.x, a0.y, a1.x, a1.y};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pv[e] = n == 0 ? h[e] : pv[e];
                const float d = h[e] - pv[e];
                s1[e] += d;
                s2[e] += d * d;
            }
            ++n;
        }
    });
    const float fn = (float)n, inv = n > 0 ? 1.f / fn : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[e * 256 + tid] = pv[e] + s1[e] * inv;                            // the thread's mean
        red[(4 + e) * 256 + tid] = fmaxf(s2[e] - s1[e] * s1[e] * inv, 0.f);   // its M2
    }
    red[8 * 256 + tid] = active ? fn : 0.f;
    __syncthreads();
    float *dst = tiles + (size_t)blockIdx.x * 3 * g.C;
    for (int c = tid; c < g.C; c += 256) {
        const int cgp = c >> 2, e = c & 3;
        double N = 0.0, S = 0.0;
        for (int q = 0; q < SL; ++q) {
            const double nq = red[8 * 256 + q * G + cgp];
            N += nq;
            S += nq * (double)red[e * 256 + q * G + cgp];
        }
        const double mu = N > 0.0 ? S / N : 0.0;
        double Q = 0.0;
        for (int q = 0; q < SL; ++q) {
            const double nq = red[8 * 256 + q * G + cgp];
            const double dl = (double)red[e * 256 + q * G + cgp] - mu;
            Q += nq > 0.0 ? (double)red[(4 + e) * 256 + q * G + cgp] + nq * dl * dl : 0.0;
        }
        dst[c] = (float)N;
        dst[g.C + c] = (float)S;
        dst[2 * g.C + c] = (float)Q;
    }
}

// y = relu((h - mean) * (invstd * gamma) + beta): bn_apply_kernel's expression on the recomputed h
template <typename TO>
__global__ __launch_bounds__(256) void c1_bn_apply_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                          const float *__restrict__ bias, const float *__restrict__ mean,
                                                          const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, TO *__restrict__ out, const C1Geom g)
{
    __shared__ __attribute__((aligned(16))) float patch[2][4 * PP];
    const int tid = threadIdx.x;
    const int G = g.C >> 2, SL = 256 / G;
    const int cg = tid % G, sl = tid / G;
    const bool active = sl < SL;
    const int c0 = cg * 4;
    v2f wp[16][2];
    load_w4(w, c0, active, wp);
    v2f bv[2] = {{0.f, 0.f}, {0.f, 0.f}};
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        if (bias) { bv[0] = v2f{bias[c0], bias[c0 + 1]}; bv[1] = v2f{bias[c0 + 2], bias[c0 + 3]}; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { mu[e] = mean[c0 + e]; sc[e] = invstd[c0 + e] * gamma[c0 + e]; be[e] = beta[c0 + e]; }
    }
    tile_loop(img, g, patch, tid, [&](int b, int ly, int ox0, int npx, const float *pt) {
        TO *orow = out + (((size_t)b * g.LH + ly) * g.LW + ox0) * g.C + c0;
        for (int j = sl; active && j < npx; j += SL) {
            v2f tp[8], a0, a1;
            read_taps(pt, j, tp);
            conv_taps(tp, wp, bv, a0, a1);
            const float h[4] = {a0.x, a0.y, a1.x, a1.y};
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf((h[e] - mu[e]) * sc[e] + be[e], 0.f);
            store4<TO>(orow + (size_t)j * g.C, y[0], y[1], y[2], y[3]);
        }
    });
}

// partial[block][2][C] = (sum g, sum g * xhat) over the block's pixels, g = dy where the ReLU passed (bn_bwd_partial_kernel's
// sums on the recomputed h); slots pooled in slot order.
template <typename TG>
__global__ __launch_bounds__(256) void c1_bn_bwd_sums_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                             const float *__restrict__ bias, const TG *__restrict__ dy,
                                                             const float *__restrict__ mean, const float *__restrict__ invstd,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float *__restrict__ partial, const C1Geom g)
{
    __shared__ __attribute__((aligned(16))) float patch[2][4 * PP];
    __shared__ float red[8 * 256];
    const int tid = threadIdx.x;
    const int G = g.C >> 2, SL = 256 / G;
    const int cg = tid % G, sl = tid / G;
    const bool active = sl < SL;
    const int c0 = cg * 4;
    v2f wp[16][2];
    load_w4(w, c0, active, wp);
    v2f bv[2] = {{0.f, 0.f}, {0.f, 0.f}};
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f}, fs[4] = {0.f, 0.f, 0.f, 0.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        if (bias) { bv[0] = v2f{bias[c0], bias[c0 + 1]}; bv[1] = v2f{bias[c0 + 2], bias[c0 + 3]}; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e]; fs[e] = is[e] * gamma[c0 + e]; be[e] = beta[c0 + e]; }
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    tile_loop(img, g, patch, tid, [&](int b, int ly, int ox0, int npx, const float *pt) {
        const TG *grow = dy + (((size_t)b * g.LH + ly) * g.LW + ox0) * g.C + c0;
        for (int j0 = sl; active && j0 < npx; j0 += 4 * SL) {
            float gv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * SL;
                load4<TG>(grow + (size_t)(j < npx ? j : j0) * g.C, gv[u]);     // clamped: always a valid address
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * SL;
                if (j < npx) {
                    v2f tp[8], a0, a1;
                    read_taps(pt, j, tp);
                    conv_taps(tp, wp, bv, a0, a1);
                    const float h[4] = {a0.x, a0.y, a1.x, a1.y};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ge = ((h[e] - mu[e]) * fs[e] + be[e]) > 0.f ? gv[u][e] : 0.f;
                        s1[e] += ge;
                        s2[e] += ge * ((h[e] - mu[e]) * is[e]);
                    }
                }
            }
        }
    });
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[e * 256 + tid] = active ? s1[e] : 0.f; red[(4 + e) * 256 + tid] = active ? s2[e] : 0.f; }
    __syncthreads();
    float *dst = partial + (size_t)blockIdx.x * 2 * g.C;
    for (int i = tid; i < 2 * g.C; i += 256) {
        const int which = i / g.C, c = i - which * g.C;
        const int cgp = c >> 2, e = c & 3;
        float t = 0.f;
        for (int q = 0; q < SL; ++q) t += red[(4 * which + e) * 256 + q * G + cgp];
        dst[i] = t;
    }
}

// dh = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) (bn_bwd_apply_kernel's expression), consumed on the spot:
// partial[block][c][17] = 16 tap sums of dh * patch + the column sum of dh (layout of c1_stencil_wgrad_kernel).
template <typename TG>
__global__ __launch_bounds__(256) void c1_bn_bwd_wgrad_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                              const float *__restrict__ bias, const TG *__restrict__ dy,
                                                              const float *__restrict__ mean, const float *__restrict__ invstd,
                                                              const float *__restrict__ gamma, const float *__restrict__ beta,
                                                              const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                              float inv_m, float *__restrict__ partial, const C1Geom g)
{
    __shared__ __attribute__((aligned(16))) float patch[2][4 * PP];
    __shared__ float red[17 * 256];
    const int tid = threadIdx.x;
    const int G = g.C >> 2, SL = 256 / G;
    const int cg = tid % G, sl = tid / G;
    const bool active = sl < SL;
    const int c0 = cg * 4;
    v2f wp[16][2];
    load_w4(w, c0, active, wp);
    v2f bv[2] = {{0.f, 0.f}, {0.f, 0.f}};
    float mu[4], is[4], fs[4], be[4], sc[4], dg[4], db[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { mu[e] = is[e] = fs[e] = be[e] = sc[e] = dg[e] = db[e] = 0.f; }
    if (active) {
        if (bias) { bv[0] = v2f{bias[c0], bias[c0 + 1]}; bv[1] = v2f{bias[c0 + 2], bias[c0 + 3]}; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            mu[e] = mean[c]; is[e] = invstd[c]; sc[e] = gamma[c] * is[e]; fs[e] = is[e] * gamma[c]; be[e] = beta[c];
            dg[e] = dgamma[c] * inv_m; db[e] = dbeta[c] * inv_m;
        }
    }
    v2f acc[4][8];
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[c][k] = v2f{0.f, 0.f};

    tile_loop(img, g, patch, tid, [&](int b, int ly, int ox0, int npx, const float *pt) {
        const TG *grow = dy + (((size_t)b * g.LH + ly) * g.LW + ox0) * g.C + c0;
        for (int j0 = sl; active && j0 < npx; j0 += 2 * SL) {
            float gv[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = j0 + u * SL;
                load4<TG>(grow + (size_t)(j < npx ? j : j0) * g.C, gv[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = j0 + u * SL;
                if (j < npx) {
                    v2f tp[8], a0, a1;
                    read_taps(pt, j, tp);
                    conv_taps(tp, wp, bv, a0, a1);
                    const float h[4] = {a0.x, a0.y, a1.x, a1.y};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float ge = ((h[c] - mu[c]) * fs[c] + be[c]) > 0.f ? gv[u][c] : 0.f;
                        const float d = sc[c] * (ge - db[c] - ((h[c] - mu[c]) * is[c]) * dg[c]);
                        const v2f dc = {d, d};
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[c][k] = __builtin_elementwise_fma(dc, tp[k], acc[c][k]);
                        cs[c] += d;
                    }
                }
            }
        }
    });
    float *dst = partial + (size_t)blockIdx.x * g.C * 17;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[(2 * k) * 256 + tid] = active ? acc[c][k].x : 0.f;
            red[(2 * k + 1) * 256 + tid] = active ? acc[c][k].y : 0.f;
        }
        red[16 * 256 + tid] = active ? cs[c] : 0.f;
        __syncthreads();
        for (int e = tid; e < G * 17; e += 256) {
            const int v = e / G, cgp = e - v * G;
            float s = 0.f;
            for (int q = 0; q < SL; ++q) s += red[v * 256 + q * G + cgp];
            dst[(size_t)(cgp * 4 + c) * 17 + v] = s;
        }
    }
}

constexpr int WGRAD_BLOCKS = 1024;

}  // namespace

bool nsg_c1_stencil_supported(int C) { return C >= 4 && C % 4 == 0 && C <= 1024; }

size_t nsg_c1_stencil_wgrad_workspace_bytes(int C) { return nsg_align_up((size_t)WGRAD_BLOCKS * C * 17 * sizeof(float), 256); }

int nsg_launch_c1_stencil_fwd(const float *img, const float *w, const float *bias, void *out, int out_dtype, int B, int LH, int LW,
                              int HH, int WW, int C, hipStream_t s)
{
    if ((int64_t)B * LH * LW <= 0) return NSG_OK;
    const C1Geom g = make_geom(B, LH, LW, HH, WW, C);
    const int blocks = g.ntiles < 2048 ? g.ntiles : 2048;
    if (out_dtype == NSG_BF16)
        hipLaunchKernelGGL((c1_stencil_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, img, w, bias, reinterpret_cast<bf16_t *>(out), g);
    else
        hipLaunchKernelGGL((c1_stencil_fwd_kernel<float>), dim3(blocks), dim3(256), 0, s, img, w, bias, reinterpret_cast<float *>(out), g);
    return nsg_check_launch("c1_stencil_fwd_kernel");
}

int nsg_launch_c1_stencil_wgrad(const float *img, const void *t, int t_dtype, int relu_t, float *dw, float *colsum, int B, int LH,
                                int LW, int HH, int WW, int C, void *ws, size_t ws_bytes, hipStream_t s)
{
    if (ws_bytes < nsg_c1_stencil_wgrad_workspace_bytes(C)) return nsg_fail(NSG_E_WORKSPACE, "c1_stencil_wgrad: workspace too small");
    const C1Geom g = make_geom(B, LH, LW, HH, WW, C);
    int blocks = g.ntiles < WGRAD_BLOCKS ? g.ntiles : WGRAD_BLOCKS;
    if (blocks < 1) blocks = 1;
    float *partial = reinterpret_cast<float *>(ws);
#define NSG_C1W(TI, R) hipLaunchKernelGGL((c1_stencil_wgrad_kernel<TI, R>), dim3(blocks), dim3(256), 0, s, img, reinterpret_cast<const TI *>(t), partial, g)
    if (t_dtype == NSG_BF16) { if (relu_t) NSG_C1W(bf16_t, true); else NSG_C1W(bf16_t, false); }
    else                     { if (relu_t) NSG_C1W(float, true);  else NSG_C1W(float, false); }
#undef NSG_C1W
    int rc = nsg_check_launch("c1_stencil_wgrad_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(c1_stencil_wgrad_final_kernel, dim3((C * 17 + 31) / 32), dim3(256), 0, s, partial, blocks, C, dw, colsum);
    return nsg_check_launch("c1_stencil_wgrad_final_kernel");
}

int nsg_launch_c1_stencil_wgrad_final(const float *partial17, int blocks, int C, float *dw, float *colsum, hipStream_t s)
{
    hipLaunchKernelGGL(c1_stencil_wgrad_final_kernel, dim3((C * 17 + 31) / 32), dim3(256), 0, s, partial17, blocks, C, dw, colsum);
    return nsg_check_launch("c1_stencil_wgrad_final_kernel");
}

// ---- the fused Conv2d(1, C, 4, 2, 1) + BatchNorm2d + ReLU layer (C ABI: include/nsg.h) ----
// c1_mfma.hip: the same four passes with the convolution on the matrix cores, for bf16 tensors
bool nsg_c1m_supported(int C);
int nsg_launch_c1m_stats(const float *img, const float *w, const float *bias, float *tiles, int blocks, int B, int LH, int LW, int HH, int WW,
                         int C, hipStream_t s);
int nsg_launch_c1m_apply(const float *img, const float *w, const float *bias, const float *mean, const float *invstd, const float *gamma,
                         const float *beta, void *out, int blocks, int B, int LH, int LW, int HH, int WW, int C, hipStream_t s);
int nsg_launch_c1m_bwd_sums(const float *img, const float *w, const float *bias, const void *dy, const float *mean, const float *invstd,
                            const float *gamma, const float *beta, float *partial, int blocks, int B, int LH, int LW, int HH, int WW, int C,
                            hipStream_t s);
int nsg_launch_c1m_bwd_wgrad(const float *img, const float *w, const float *bias, const void *dy, const float *mean, const float *invstd,
                             const float *gamma, const float *beta, const float *dgamma, const float *dbeta, float inv_m, float *partial,
                             int blocks, int B, int LH, int LW, int HH, int WW, int C, hipStream_t s);

// ... and the input layer by its tap moments (c1_mfma.hip): statistics without a pass over h, backward in one pass over dy
size_t nsg_c1m_moments_bytes();
int nsg_launch_c1m_moments(const float *img, int B, int LH, int LW, int HH, int WW, void *ws, double *mom_dst, const double **mom_out, hipStream_t s);
int nsg_launch_c1m_stats_from_moments(const double *mom, const float *w, const float *bias, int64_t M, int C, float eps, float momentum,
                                      float *mean, float *invstd, float *running_mean, float *running_var, hipStream_t s);
int nsg_launch_c1m_bwd_onepass(const float *img, const float *w, const float *bias, const void *dy, const float *mean, const float *invstd,
                               const float *gamma, const float *beta, float *sums, float *partial17, int blocks, int B, int LH, int LW,
                               int HH, int WW, int C, hipStream_t s);
int nsg_launch_c1m_onepass_fixup(const double *mom, const float *w, const float *bias, const float *mean, const float *invstd,
                                 const float *gamma, const float *dgamma, const float *dbeta, int64_t M, int C, float *dw, float *dbias,
                                 hipStream_t s);
NSG_DIAG_SWITCH(int, g_c1_moments, 1)     // nsg_debug_set_c1_moments (diagnostics library only): 0 = the two-pass forms (statistics pass over h; sums pass + gradient pass over dy)
#ifdef NSG_DIAG
extern "C" NSG_API void nsg_debug_set_c1_moments(int on) { g_c1_moments = on; }
#endif

namespace {
constexpr int FUSED_BLOCKS = 1024;      // = bn.hip's MAX_SLABS (bn_bwd_final_kernel) and WGRAD_BLOCKS
size_t fused_tiles_bytes(int C) { return nsg_align_up(nsg_bn_tiles_bytes(2 * FUSED_BLOCKS, C), 256); }
size_t fused_sums_bytes(int C) { return nsg_align_up((size_t)FUSED_BLOCKS * 2 * C * sizeof(float), 256); }
int check_c1bn(const char *fn, int B, int H, int W, int C)
{
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return nsg_fail(NSG_E_INVALID, "%s: the image extent must be positive and even (H=%d, W=%d)", fn, H, W);
    if (!nsg_c1_stencil_supported(C)) return nsg_fail(NSG_E_UNSUPPORTED, "%s: C=%d must be a multiple of 4 in [4, 1024]", fn, C);
    if ((int64_t)B * (H / 2) * (W / 2) * C > 0x7fffffffLL * 4) return nsg_fail(NSG_E_UNSUPPORTED, "%s: tensor too large", fn);
    return NSG_OK;
}
}  // namespace

extern "C" {

size_t nsg_c1conv_bn_workspace_bytes(int32_t C)
{
    if (C <= 0) return 0;
    const size_t fwd = fused_tiles_bytes(C);
    const size_t bwd = fused_sums_bytes(C) + nsg_align_up(nsg_c1_stencil_wgrad_workspace_bytes(C), 256);
    return (fwd > bwd ? fwd : bwd) + nsg_c1m_moments_bytes();       // (the moments sit behind the larger of the two)
}

int nsg_c1conv_bn_relu_forward(const float *img, const float *w, const float *bias, const float *gamma, const float *beta, float *mean,
                               float *invstd, float *running_mean, float *running_var, float eps, float momentum, int32_t training,
                               void *y, int32_t y_dtype, int32_t B, int32_t H, int32_t W, int32_t C, void *workspace,
                               size_t workspace_bytes, double *moments, void *stream)
{
    NSG_REQUIRE(img && w && gamma && beta && mean && invstd && y, NSG_E_INVALID, "nsg_c1conv_bn_relu_forward: null pointer");
    NSG_REQUIRE(y_dtype == NSG_F32 || y_dtype == NSG_BF16, NSG_E_INVALID, "nsg_c1conv_bn_relu_forward: y_dtype must be NSG_F32 or NSG_BF16");
    int rc = check_c1bn("nsg_c1conv_bn_relu_forward", B, H, W, C);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(w) && nsg_aligned16(y), NSG_E_INVALID, "nsg_c1conv_bn_relu_forward: w and y must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const C1Geom g = make_geom(B, H / 2, W / 2, H, W, C);
    // bf16 output: the passes with the convolution on the matrix cores (c1_mfma.hip); the statistics then come from the
    // same instruction sequence as the values they normalise
    const bool mfma = y_dtype == NSG_BF16 && nsg_c1m_supported(C);
    if (training) {
        NSG_REQUIRE(workspace && workspace_bytes >= nsg_c1conv_bn_workspace_bytes(C), NSG_E_WORKSPACE, "nsg_c1conv_bn_relu_forward: workspace too small");
        const int blocks = g.ntiles < 2 * FUSED_BLOCKS ? g.ntiles : 2 * FUSED_BLOCKS;
        float *tiles = reinterpret_cast<float *>(workspace);
        if (mfma && g_c1_moments) {      // statistics of h from the image's tap moments: no pass over h
            const double *mom = nullptr;
            rc = nsg_launch_c1m_moments(img, B, g.LH, g.LW, H, W, reinterpret_cast<char *>(workspace) + nsg_c1conv_bn_workspace_bytes(C) - nsg_c1m_moments_bytes(), moments, &mom, s);
            if (rc) return rc;
            rc = nsg_launch_c1m_stats_from_moments(mom, w, bias, (int64_t)B * g.LH * g.LW, C, eps, momentum, mean, invstd, running_mean, running_var, s);
            if (rc) return rc;
            const int ablocks = g.ntiles < 2048 ? g.ntiles : 2048;
            return nsg_launch_c1m_apply(img, w, bias, mean, invstd, gamma, beta, y, ablocks, B, g.LH, g.LW, H, W, C, s);
        }
        if (mfma) {
            rc = nsg_launch_c1m_stats(img, w, bias, tiles, blocks, B, g.LH, g.LW, H, W, C, s);
        } else {
            hipLaunchKernelGGL(c1_bn_stats_kernel, dim3(blocks), dim3(256), 0, s, img, w, bias, tiles, g);
            rc = nsg_check_launch("c1_bn_stats_kernel");
        }
        if (rc) return rc;
        rc = nsg_bn_stats_from_tiles(tiles, blocks, (int64_t)B * g.LH * g.LW, C, eps, momentum, mean, invstd, running_mean, running_var, s);
        if (rc) return rc;
    }
    const int blocks = g.ntiles < 2048 ? g.ntiles : 2048;
    if (mfma) return nsg_launch_c1m_apply(img, w, bias, mean, invstd, gamma, beta, y, blocks, B, g.LH, g.LW, H, W, C, s);
    if (y_dtype == NSG_BF16)
        hipLaunchKernelGGL((c1_bn_apply_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, img, w, bias, mean, invstd, gamma, beta, reinterpret_cast<bf16_t *>(y), g);
    else
        hipLaunchKernelGGL((c1_bn_apply_kernel<float>), dim3(blocks), dim3(256), 0, s, img, w, bias, mean, invstd, gamma, beta, reinterpret_cast<float *>(y), g);
    return nsg_check_launch("c1_bn_apply_kernel");
}

int nsg_c1conv_bn_relu_backward(const float *img, const float *w, const float *bias, const float *gamma, const float *beta,
                                const float *mean, const float *invstd, const void *dy, int32_t dy_dtype, float *dw, float *dbias,
                                float *dgamma, float *dbeta, int32_t B, int32_t H, int32_t W, int32_t C, void *workspace,
                                size_t workspace_bytes, const double *moments, void *stream)
{
    NSG_REQUIRE(img && w && gamma && beta && mean && invstd && dy && dw && dgamma && dbeta, NSG_E_INVALID, "nsg_c1conv_bn_relu_backward: null pointer");
    NSG_REQUIRE(dy_dtype == NSG_F32 || dy_dtype == NSG_BF16, NSG_E_INVALID, "nsg_c1conv_bn_relu_backward: dy_dtype must be NSG_F32 or NSG_BF16");
    int rc = check_c1bn("nsg_c1conv_bn_relu_backward", B, H, W, C);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(w) && nsg_aligned16(dy), NSG_E_INVALID, "nsg_c1conv_bn_relu_backward: w and dy must be 16-byte aligned");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_c1conv_bn_workspace_bytes(C), NSG_E_WORKSPACE, "nsg_c1conv_bn_relu_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const C1Geom g = make_geom(B, H / 2, W / 2, H, W, C);
    const int blocks = g.ntiles < FUSED_BLOCKS ? g.ntiles : FUSED_BLOCKS;
    float *sums = reinterpret_cast<float *>(workspace);
    float *partial17 = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + fused_sums_bytes(C));
    const float inv_m = 1.f / (float)((int64_t)B * g.LH * g.LW);
    if (dy_dtype == NSG_BF16 && nsg_c1m_supported(C) && g_c1_moments) {      // one pass over dy (c1_mfma.hip: "by its tap moments")
        const double *mom = moments;
        const int64_t M = (int64_t)B * g.LH * g.LW;
        if (!mom) {
            rc = nsg_launch_c1m_moments(img, B, g.LH, g.LW, H, W, reinterpret_cast<char *>(workspace) + nsg_c1conv_bn_workspace_bytes(C) - nsg_c1m_moments_bytes(), nullptr, &mom, s);
            if (rc) return rc;
        }
        rc = nsg_launch_c1m_bwd_onepass(img, w, bias, dy, mean, invstd, gamma, beta, sums, partial17, blocks, B, g.LH, g.LW, H, W, C, s);
        if (rc) return rc;
        rc = nsg_launch_bn_bwd_final(sums, blocks, C, dgamma, dbeta, s);
        if (rc) return rc;
        hipLaunchKernelGGL(c1_stencil_wgrad_final_kernel, dim3((C * 17 + 31) / 32), dim3(256), 0, s, partial17, blocks, C, dw, (float *)nullptr);
        rc = nsg_check_launch("c1_stencil_wgrad_final_kernel");
        if (rc) return rc;
        return nsg_launch_c1m_onepass_fixup(mom, w, bias, mean, invstd, gamma, dgamma, dbeta, M, C, dw, dbias, s);
    }
    if (dy_dtype == NSG_BF16 && nsg_c1m_supported(C)) {     // the forward's choice for bf16 tensors: the same h, the same ReLU decisions
        rc = nsg_launch_c1m_bwd_sums(img, w, bias, dy, mean, invstd, gamma, beta, sums, blocks, B, g.LH, g.LW, H, W, C, s);
        if (rc) return rc;
        rc = nsg_launch_bn_bwd_final(sums, blocks, C, dgamma, dbeta, s);
        if (rc) return rc;
        rc = nsg_launch_c1m_bwd_wgrad(img, w, bias, dy, mean, invstd, gamma, beta, dgamma, dbeta, inv_m, partial17, blocks, B, g.LH, g.LW, H, W, C, s);
        if (rc) return rc;
        hipLaunchKernelGGL(c1_stencil_wgrad_final_kernel, dim3((C * 17 + 31) / 32), dim3(256), 0, s, partial17, blocks, C, dw, dbias);
        return nsg_check_launch("c1_stencil_wgrad_final_kernel");
    }
    if (dy_dtype == NSG_BF16)
        hipLaunchKernelGGL((c1_bn_bwd_sums_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, img, w, bias, reinterpret_cast<const bf16_t *>(dy), mean, invstd, gamma, beta, sums, g);
    else
        hipLaunchKernelGGL((c1_bn_bwd_sums_kernel<float>), dim3(blocks), dim3(256), 0, s, img, w, bias, reinterpret_cast<const float *>(dy), mean, invstd, gamma, beta, sums, g);
    rc = nsg_check_launch("c1_bn_bwd_sums_kernel");
    if (rc) return rc;
    rc = nsg_launch_bn_bwd_final(sums, blocks, C, dgamma, dbeta, s);
    if (rc) return rc;
    if (dy_dtype == NSG_BF16)
        hipLaunchKernelGGL((c1_bn_bwd_wgrad_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, img, w, bias, reinterpret_cast<const bf16_t *>(dy), mean, invstd, gamma, beta,
                           dgamma, dbeta, inv_m, partial17, g);
    else
        hipLaunchKernelGGL((c1_bn_bwd_wgrad_kernel<float>), dim3(blocks), dim3(256), 0, s, img, w, bias, reinterpret_cast<const float *>(dy), mean, invstd, gamma, beta,
                           dgamma, dbeta, inv_m, partial17, g);
    rc = nsg_check_launch("c1_bn_bwd_wgrad_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(c1_stencil_wgrad_final_kernel, dim3((C * 17 + 31) / 32), dim3(256), 0, s, partial17, blocks, C, dw, dbias);
    return nsg_check_launch("c1_stencil_wgrad_final_kernel");
}

}  // extern "C"
