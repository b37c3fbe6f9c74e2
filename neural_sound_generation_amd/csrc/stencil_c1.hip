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

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int TW = 64;             // low-res pixels of one image row per tile
constexpr int PP = 2 * TW + 4;     // patch row pitch (2*TW + 2 used)

struct C1Geom {
    int B, LH, LW, HH, WW, C;
    int segs;                      // tiles per low-res row
    int ntiles;
    FastDiv div_segs, div_lh;
};

// The 4 image rows x (2*TW+2) columns a tile needs (out-of-image positions are zero: the conv padding), in two halves so
// that the NEXT tile's global loads are in flight while the current tile is computed: patch_load -> registers,
// patch_store -> LDS.  4*(2*TW+2) = 520 values = 3 per thread of 256 (the last partly).
constexpr int PATCH_VALS = 4 * (2 * TW + 2);
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

__device__ __forceinline__ void tile_coords(const C1Geom &g, int tile, int &b, int &ly, int &ox0)
{
    const int row = nsg_div(tile, g.div_segs);
    ox0 = (tile - row * g.segs) * TW;
    b = nsg_div(row, g.div_lh);
    ly = row - b * g.LH;
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

    // weights of this thread's 4 channels as channel pairs: wp[t][0] = (w[c0][t], w[c0+1][t]), wp[t][1] = (w[c0+2][t], w[c0+3][t])
    v2f wp[16][2];
    v2f bv[2] = {{0.f, 0.f}, {0.f, 0.f}};
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
        if (bias) { bv[0] = v2f{bias[c0], bias[c0 + 1]}; bv[1] = v2f{bias[c0 + 2], bias[c0 + 3]}; }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) { wp[t][0] = v2f{0.f, 0.f}; wp[t][1] = v2f{0.f, 0.f}; }
    }

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
            v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
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
            TO *dst = orow + (size_t)j * g.C;
            if constexpr (sizeof(TO) == 4) {
                *reinterpret_cast<v4f *>(dst) = v4f{a0.x, a0.y, a1.x, a1.y};
            } else {
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const v2u pk = {(unsigned)nsg_f2bf(a0.x) | ((unsigned)nsg_f2bf(a0.y) << 16),
                                (unsigned)nsg_f2bf(a1.x) | ((unsigned)nsg_f2bf(a1.y) << 16)};
                *reinterpret_cast<v2u *>(dst) = pk;
            }
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

C1Geom make_geom(int B, int LH, int LW, int HH, int WW, int C)
{
    C1Geom g;
    g.B = B; g.LH = LH; g.LW = LW; g.HH = HH; g.WW = WW; g.C = C;
    g.segs = (LW + TW - 1) / TW;
    g.ntiles = B * LH * g.segs;
    g.div_segs = nsg_fastdiv((uint32_t)g.segs);
    g.div_lh = nsg_fastdiv((uint32_t)LH);
    return g;
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
