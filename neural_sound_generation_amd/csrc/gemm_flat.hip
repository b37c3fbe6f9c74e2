// The ResBlock's 1x1 convolution (src/models.py:153, Conv2d(dim, dim, 1)) with the BatchNorm work of its neighbours folded
// into its operand staging, bf16 mode.  A 1x1 conv over NHWC rows is a flat GEMM out[M][C] = in[M][C] * W^T with K = C <= 128:
// 2C flops per byte moved, i.e. purely a stream kernel -- the gather / index machinery of gather_gemm buys nothing here
// and its fixed per-tile cost is 70% of the time.  What matters is how many times the [M][C] tensors cross HBM:
//   forward   y  = relu(bn1(x)) * W^T + b        the activated tensor a = relu(bn1(x)) is built on the way from the
//                                                 global-load registers into LDS and never stored
//             (nsg_bn_apply + nsg_conv_forward: read x, write a, read a, write y -> read x, write y)
//   backward  dh = bn2-backward(dy, h)            built the same way, stored once (the weight gradient needs it), and
//             dx = dh * W                          multiplied on the spot; the column sums of dh (the conv bias gradient) fall out
//             (nsg_bn_backward's apply pass + nsg_conv_dgrad: read dy, h, write dh, read dh, write dx -> read dy, h, write dh, dx)
// Block = 4 waves = one 128-row tile at a time (persistent); wave w owns rows 32w..32w+31.  MFMA (32x32x16 bf16):
// out^T[n][row] = W[n][k] * in^T[k][row]: both operands are 16-byte row-major LDS reads (lane = W row / tile row, 8
// consecutive k).  A lane ends up with 4 consecutive output channels of its row per register group; the tile is
// rewritten in place in LDS (a wave only ever touches its own rows) and leaves as 16-byte coalesced stores.
#include "nsg_common.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const v2f f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

constexpr int ROWS = 128;

struct FlatParams {
    const bf16_t *x;        // FWD: the BatchNorm input;  BWD: the BatchNorm input h (of the BatchNorm being back-propagated)
    const bf16_t *g;        // BWD: dy
    const float *w;         // (C, C, 1, 1) = [out][in] fp32
    const float *bias;      // FWD: [C] or null
    const float *mean, *invstd, *gamma, *beta;   // the BatchNorm's statistics and parameters [C] (beta: FWD only)
    const float *dgamma, *dbeta;                 // BWD: the BatchNorm's parameter gradients (nsg_bn_backward_sums)
    float inv_m;                                 // BWD: 1 / M
    bf16_t *out;            // FWD: y;  BWD: dx
    bf16_t *mid;            // BWD: dh
    float *colsum_partial;  // BWD: [gridDim.x][C] column sums of dh over the block's tiles (or null)
    // BWD, optional: the sums of the BatchNorm + ReLU that sits in FRONT of the conv (input prev_x, output = the conv's input):
    // prev_partial[gridDim.x][2][C] = (sum g, sum g * xhat) with g = dx where that ReLU passed -- nsg_bn_backward_sums of
    // (prev_x, dx) formed while dx is written
    const bf16_t *prev_x;
    const float *prev_mean, *prev_invstd, *prev_gamma, *prev_beta;
    float *prev_partial;
    float *stat_tiles;      // FWD: [gridDim.x][3][C] (count, sum, M2 about the block mean) of the stored output y (or null):
                            //      the batch statistics of the BatchNorm that follows, from the store phase instead of a pass over y
    int64_t M;
};

// MODE 0: forward (a = max(fma(x, fs, off), 0));  MODE 1: backward (dh = fma(sc, dy, -fma(k1, h, k0)); W used transposed)
template <int NB, int MODE>     // C = 32 NB
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void flat_gemm_kernel(const FlatParams p)
{
    constexpr int C = 32 * NB, KS = C / 16, PITCH = C + 8, CPR = C / 8, PIECES = ROWS * CPR / 256;
    static_assert(256 % CPR == 0, "a thread stages the same channel group in every piece");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *tile = reinterpret_cast<bf16_t *>(smem_raw);          // [ROWS][PITCH]
    bf16_t *wt = tile + ROWS * PITCH;                             // [C (n)][PITCH (k)]
    __shared__ __attribute__((aligned(16))) float sbias[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hb = lane >> 5;

    // ---- weights -> LDS as bf16 [n][k]: FWD n = out channel, k = in channel (w[n][k]); BWD n = in channel, k = out channel (w[k][n]) ----
    // (all the loads first, then the conversions: a rolled loop waits for every load in turn, 32 L2 round trips per block)
    constexpr int WIT = C * C / 2 / 256;
    {
        float wa[WIT], wb[WIT];
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + 256 * it;                       // FWD: lanes along k (a row of w);  BWD: lanes along n (rows k, k + 1 of w)
            const int n = MODE == 0 ? i / (C / 2) : i % C, k = MODE == 0 ? (i - n * (C / 2)) * 2 : (i / C) * 2;
            wa[it] = MODE == 0 ? p.w[(size_t)n * C + k] : p.w[(size_t)k * C + n];
            wb[it] = MODE == 0 ? p.w[(size_t)n * C + k + 1] : p.w[(size_t)(k + 1) * C + n];
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + 256 * it;
            const int n = MODE == 0 ? i / (C / 2) : i % C, k = MODE == 0 ? (i - n * (C / 2)) * 2 : (i / C) * 2;
            *reinterpret_cast<unsigned *>(wt + n * PITCH + k) = pack_bf16(wa[it], wb[it]);
        }
    }
    if (tid < C) sbias[tid] = (MODE == 0 && p.bias) ? p.bias[tid] : 0.f;
    // ---- this thread's staging pieces: piece i = (row tid / CPR + (256 / CPR) i, channels cc8 .. cc8+7), the SAME channels for every piece ----
    const int prow = tid / CPR, cc8 = (tid % CPR) * 8;
    constexpr int RSTEP = 256 / CPR;
    // per-channel constants of the staging transform.  FWD: k0 = fs = invstd*gamma, k1 = off = beta - mean*fs.
    // BWD: k0 = sc = gamma*invstd, k1 = sc*invstd*dgamma/M, k2 = sc*dbeta/M - k1*mean  (dh = sc*dy - (k1*h + k2))
    // (BWD keeps them in LDS and reads them at the start of the phase that uses them: the kernel sits at the 256-register limit)
    __shared__ __attribute__((aligned(16))) float sconst[MODE == 1 ? 6 * C : 4];
    float k0[8], k1[8], k2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cc8 + e;
        if (MODE == 0) {
            k0[e] = p.invstd[c] * p.gamma[c];
            k1[e] = __builtin_fmaf(-p.mean[c], k0[e], p.beta[c]);
            k2[e] = 0.f;
        } else {
            k0[e] = p.gamma[c] * p.invstd[c];
            k1[e] = k0[e] * p.invstd[c] * (p.dgamma[c] * p.inv_m);
            k2[e] = __builtin_fmaf(k0[e], p.dbeta[c] * p.inv_m, -(k1[e] * p.mean[c]));
        }
    }
    float csum[8], ssq[8], spv[8], scnt = 0.f;     // BWD: column sums of dh.  FWD: sums of (y - pivot), (y - pivot)^2, the pivot, the count
#pragma unroll
    for (int e = 0; e < 8; ++e) { csum[e] = 0.f; ssq[e] = 0.f; spv[e] = 0.f; }
    const bool prev = MODE == 1 && p.prev_x != nullptr;
    float pmu[8], pfs[8], pbe[8], ps1[8], ps2[8];     // the BatchNorm in front: mean, invstd*gamma, beta; its two sums
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        pmu[e] = prev ? p.prev_mean[cc8 + e] : 0.f;
        pfs[e] = prev ? p.prev_invstd[cc8 + e] * p.prev_gamma[cc8 + e] : 0.f;
        pbe[e] = prev ? p.prev_beta[cc8 + e] : 0.f;
        ps1[e] = 0.f; ps2[e] = 0.f;
    }
    if (MODE == 1 && prow == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sconst[cc8 + e] = k0[e]; sconst[C + cc8 + e] = k1[e]; sconst[2 * C + cc8 + e] = k2[e];
            sconst[3 * C + cc8 + e] = pmu[e]; sconst[4 * C + cc8 + e] = pfs[e]; sconst[5 * C + cc8 + e] = pbe[e];
        }
    }
    if (MODE == 1) __syncthreads();
    auto consts = [&](int j, float (&v)[8]) __attribute__((always_inline)) {
        const v4f a = *reinterpret_cast<const v4f *>(sconst + j * C + cc8), b = *reinterpret_cast<const v4f *>(sconst + j * C + cc8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    };
    const int64_t ntiles = (p.M + ROWS - 1) / ROWS;
    // FWD keeps TWO tiles of loads in flight (register sets px and py, used alternately); BWD one (it also carries dy and prev_x)
    v4f px[PIECES], py[MODE == 0 ? PIECES : 1], pg[MODE == 1 ? PIECES : 1];
    auto prefetch = [&](int64_t t, v4f (&px)[PIECES]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int64_t m = t * ROWS + prow + RSTEP * i;
            const size_t off = (size_t)(m < p.M ? m : p.M - 1) * C + cc8;     // clamped: always inside the tensor
            px[i] = *reinterpret_cast<const v4f *>(p.x + off);
            if (MODE == 1) pg[i] = *reinterpret_cast<const v4f *>(p.g + off);
        }
    };
    auto stage = [&](int64_t t, v4f (&px)[PIECES]) __attribute__((always_inline)) {       // registers -> transform -> LDS tile (and, backward, dh -> global)
        if (MODE == 1) { consts(0, k0); consts(1, k1); consts(2, k2); }
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int row = prow + RSTEP * i;
            const int64_t m = t * ROWS + row;
            float xv[8], o[8];
            Elem<bf16_t>::unpack16(px[i], xv);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaxf(__builtin_fmaf(xv[e], k0[e], k1[e]), 0.f);
            } else {
                float gv[8];
                Elem<bf16_t>::unpack16(pg[i], gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(k0[e], gv[e], -__builtin_fmaf(k1[e], xv[e], k2[e]));
            }
            const v4u pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
            *reinterpret_cast<v4u *>(tile + row * PITCH + cc8) = pk;
            if (MODE == 1 && m < p.M) {
                *reinterpret_cast<v4u *>(p.mid + (size_t)m * C + cc8) = pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[e] += o[e];
            }
        }
    };

    const int64_t G = gridDim.x;
    if ((int64_t)blockIdx.x < ntiles) { prefetch(blockIdx.x, px); stage(blockIdx.x, px); }
    if constexpr (MODE == 0) {      // (rows past the end are clamped, so these loads and the staging below need no conditions: no branches
        prefetch(blockIdx.x + G, px);                      // between a load and its use, which hipcc answers with s_waitcnt vmcnt(0))
        prefetch(blockIdx.x + 2 * G, py);
    }
    __syncthreads();
    auto one_tile = [&](int64_t t, v4f (&pn)[PIECES]) __attribute__((always_inline)) {     // pn: FWD the set holding tile t + G
        const int64_t nxt = t + G;
        if (MODE == 1 && nxt < ntiles) prefetch(nxt, pn);
        v4f ph[MODE == 1 ? PIECES : 1];
        if (prev) {                            // the rows of prev_x under THIS tile: in flight across the MFMA loop, used in the store phase
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int64_t m = t * ROWS + prow + RSTEP * i;
                ph[i] = *reinterpret_cast<const v4f *>(p.prev_x + (size_t)(m < p.M ? m : p.M - 1) * C + cc8);
            }
        }
        v16f acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const bf16_t *brow = tile + (32 * wave + l31) * PITCH + 8 * hb;
        const bf16_t *arow = wt + l31 * PITCH + 8 * hb;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4f *>(brow + 16 * ks));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4f *>(arow + 32 * nb * PITCH + 16 * ks));
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nb], 0, 0, 0);
            }
        }
        // result of this wave's 32 rows back over its own rows of the tile (no other wave reads them): lane = row,
        // registers 4q .. 4q+3 of block nb = channels 32 nb + 8 q + 4 hb .. +3
        bf16_t *orow = tile + (32 * wave + l31) * PITCH;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = 32 * nb + 8 * q + 4 * hb;
                const v4f bv = *reinterpret_cast<const v4f *>(sbias + ch);
                const v2u pk = {pack_bf16(acc[nb][4 * q] + bv.x, acc[nb][4 * q + 1] + bv.y), pack_bf16(acc[nb][4 * q + 2] + bv.z, acc[nb][4 * q + 3] + bv.w)};
                *reinterpret_cast<v2u *>(orow + ch) = pk;
            }
        __syncthreads();                       // the output tile is complete
        if (prev) { consts(3, pmu); consts(4, pfs); consts(5, pbe); }
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int row = prow + RSTEP * i;
            const int64_t m = t * ROWS + row;
            if (m < p.M) {
                const v4f piece = *reinterpret_cast<const v4f *>(tile + row * PITCH + cc8);
                *reinterpret_cast<v4f *>(p.out + (size_t)m * C + cc8) = piece;
                if (prev) {                    // bn_bwd_partial_kernel's sums on (prev_x, the dx values as stored), same mask expression
                    float dv[8], hv[8];
                    Elem<bf16_t>::unpack16(piece, dv);
                    Elem<bf16_t>::unpack16(ph[i], hv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float hc = hv[e] - pmu[e];
                        const float ge = (hc * pfs[e] + pbe[e]) > 0.f ? dv[e] : 0.f;
                        ps1[e] += ge;
                        ps2[e] = __builtin_fmaf(ge, hc, ps2[e]);
                    }
                }
                if (MODE == 0 && p.stat_tiles) {          // one pass about a pivot (the thread's first value), of the values as stored
                    float yv[8];
                    Elem<bf16_t>::unpack16(piece, yv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        spv[e] = scnt == 0.f ? yv[e] : spv[e];
                        const float d = yv[e] - spv[e];
                        csum[e] += d;
                        ssq[e] = __builtin_fmaf(d, d, ssq[e]);
                    }
                    scnt += 1.f;
                }
            }
        }
        __syncthreads();                       // ... and has left LDS
        if (MODE == 0) {
            stage(nxt, pn);
            prefetch(t + 3 * G, pn);
        } else if (nxt < ntiles) {
            stage(nxt, pn);
        }
        __syncthreads();
    };
    if constexpr (MODE == 0) {
        for (int64_t t = blockIdx.x; t < ntiles; t += 2 * G) {
            one_tile(t, px);
            if (t + G < ntiles) one_tile(t + G, py);
        }
    } else {
        for (int64_t t = blockIdx.x; t < ntiles; t += G) one_tile(t, px);
    }
    if (MODE == 0 && p.stat_tiles) {
        // (count, mean, M2) of each thread, pooled over the 256 / CPR threads of its channel group in row-slot order (double)
        float *red = reinterpret_cast<float *>(tile);      // [17][256]: 8 means, 8 M2s, the count
        const float inv = scnt > 0.f ? 1.f / scnt : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[e * 256 + tid] = spv[e] + csum[e] * inv;
            red[(8 + e) * 256 + tid] = fmaxf(ssq[e] - csum[e] * csum[e] * inv, 0.f);
        }
        red[16 * 256 + tid] = scnt;
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            const int grp = c >> 3, e = c & 7;
            double N = 0.0, S = 0.0;
            for (int r = 0; r < RSTEP; ++r) {
                const double n = red[16 * 256 + r * CPR + grp];
                N += n;
                S += n * (double)red[e * 256 + r * CPR + grp];
            }
            const double mu = N > 0.0 ? S / N : 0.0;
            double Q = 0.0;
            for (int r = 0; r < RSTEP; ++r) {
                const double n = red[16 * 256 + r * CPR + grp];
                const double dl = (double)red[e * 256 + r * CPR + grp] - mu;
                Q += n > 0.0 ? (double)red[(8 + e) * 256 + r * CPR + grp] + n * dl * dl : 0.0;
            }
            float *dst = p.stat_tiles + (size_t)blockIdx.x * 3 * C;
            dst[c] = (float)N;
            dst[C + c] = (float)S;
            dst[2 * C + c] = (float)Q;
        }
    }
    if (prev) {
        float *red = reinterpret_cast<float *>(tile);      // [16][256]
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[e * 256 + tid] = ps1[e]; red[(8 + e) * 256 + tid] = ps2[e]; }
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            const int grp = c >> 3, e = c & 7;
            float a = 0.f, b = 0.f;
            for (int r = 0; r < RSTEP; ++r) { a += red[e * 256 + r * CPR + grp]; b += red[(8 + e) * 256 + r * CPR + grp]; }
            p.prev_partial[(size_t)blockIdx.x * 2 * C + c] = a;
            p.prev_partial[(size_t)blockIdx.x * 2 * C + C + c] = b * p.prev_invstd[c];
        }
        __syncthreads();
    }
    if (MODE == 1 && p.colsum_partial) {
        // column sums of dh over this block's tiles: threads with the same channel group (tid % CPR) are combined in row-slot order
        float *red = reinterpret_cast<float *>(tile);      // the tile is free now (the loop ended on a barrier)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = csum[e];
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            const int grp = c >> 3, e = c & 7;
            float s = 0.f;
            for (int r = 0; r < RSTEP; ++r) s += red[(r * CPR + grp) * 8 + e];
            p.colsum_partial[(size_t)blockIdx.x * C + c] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// C = 256 (BASELINE configs[3]).  The [256][256] bf16 weights are 128 KB: next to a row tile they do not fit in LDS, but they
// fit in REGISTERS -- 8 waves, wave w keeps the 32 output channels 32w .. 32w+31 (all 256 k: 16 fragments = 64 VGPRs) for the
// whole persistent block and multiplies them with EVERY row of the tile.  LDS holds only rows: two input tiles of 64 rows
// (the next one is staged while this one is multiplied) and one output tile; 2 barriers per tile.  1 block of 512 threads
// per CU; per tile 64 KB cross HBM, ~2 K clocks of MFMA per SIMD and ~2 K clocks of LDS reads against ~8 K clocks of HBM time.
// Same arithmetic, rounding points and reduction shapes as flat_gemm_kernel (the staging transform, the bf16 store, the
// per-thread pivot sums of the statistics), so the two give the same results where both apply.
// ------------------------------------------------------------------------------------------------
constexpr int WROWS = 64;

template <int MODE>
__global__ __launch_bounds__(512) void flat_wide_kernel(const FlatParams p)
{
    constexpr int C = 256, NT = 512, KS = C / 16, PITCH = C + 8, CPR = C / 8, RSTEP = NT / CPR, PIECES = WROWS / RSTEP;
    static_assert(PIECES * RSTEP == WROWS, "whole pieces");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *tin = reinterpret_cast<bf16_t *>(smem_raw);           // [2][WROWS][PITCH]
    bf16_t *tout = tin + 2 * WROWS * PITCH;                       // [WROWS][PITCH]
    __shared__ __attribute__((aligned(16))) float sconst[6 * C];  // FWD: fs, off;  BWD: sc, k1, k2, and the BatchNorm in front: mean, fs, beta
    __shared__ __attribute__((aligned(16))) float sbias[C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hb = lane >> 5;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));

    // ---- this wave's weights -> registers: A[n][k], n = 32 wave + l31, k = 16 ks + 8 hb .. +7.  FWD: A = w[n][k];  BWD: A = w[k][n] ----
    bf16x8 wreg[KS];
    {
        const int n = 32 * wave + l31;
        if (MODE == 0) {
            v4f lo[KS], hi[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                lo[ks] = *reinterpret_cast<const v4f *>(p.w + (size_t)n * C + 16 * ks + 8 * hb);
                hi[ks] = *reinterpret_cast<const v4f *>(p.w + (size_t)n * C + 16 * ks + 8 * hb + 4);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4u pk = {pack_bf16(lo[ks].x, lo[ks].y), pack_bf16(lo[ks].z, lo[ks].w), pack_bf16(hi[ks].x, hi[ks].y), pack_bf16(hi[ks].z, hi[ks].w)};
                wreg[ks] = __builtin_bit_cast(bf16x8, pk);
            }
        } else {
#pragma unroll
            for (int half = 0; half < 2; ++half) {      // (8 k-steps at a time: 64 loads in flight per lane)
                float f[KS / 2][8];
#pragma unroll
                for (int ks = 0; ks < KS / 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[ks][e] = p.w[(size_t)(16 * (ks + half * KS / 2) + 8 * hb + e) * C + n];
#pragma unroll
                for (int ks = 0; ks < KS / 2; ++ks) {
                    const v4u pk = {pack_bf16(f[ks][0], f[ks][1]), pack_bf16(f[ks][2], f[ks][3]), pack_bf16(f[ks][4], f[ks][5]), pack_bf16(f[ks][6], f[ks][7])};
                    wreg[ks + half * KS / 2] = __builtin_bit_cast(bf16x8, pk);
                }
            }
        }
    }
    // ---- per-channel constants of the staging transform and of the store phase -> LDS (read at the start of the phase using them) ----
    const bool prev = MODE == 1 && p.prev_x != nullptr;
    if (tid < C) {
        const int c = tid;
        if (MODE == 0) {
            const float fs = p.invstd[c] * p.gamma[c];
            sconst[c] = fs;
            sconst[C + c] = __builtin_fmaf(-p.mean[c], fs, p.beta[c]);
            sbias[c] = p.bias ? p.bias[c] : 0.f;
        } else {
            const float sc = p.gamma[c] * p.invstd[c];
            const float k1 = sc * p.invstd[c] * (p.dgamma[c] * p.inv_m);
            sconst[c] = sc;
            sconst[C + c] = k1;
            sconst[2 * C + c] = __builtin_fmaf(sc, p.dbeta[c] * p.inv_m, -(k1 * p.mean[c]));
            sconst[3 * C + c] = prev ? p.prev_mean[c] : 0.f;
            sconst[4 * C + c] = prev ? p.prev_invstd[c] * p.prev_gamma[c] : 0.f;
            sconst[5 * C + c] = prev ? p.prev_beta[c] : 0.f;
            sbias[c] = 0.f;
        }
    }
    __syncthreads();
    const int prow = tid / CPR, cc8 = (tid % CPR) * 8;
    auto consts = [&](int j, float (&v)[8]) __attribute__((always_inline)) {
        const v4f a = *reinterpret_cast<const v4f *>(sconst + j * C + cc8), b = *reinterpret_cast<const v4f *>(sconst + j * C + cc8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    };
    float csum[8], ssq[8], spv[8], scnt = 0.f;     // BWD: column sums of dh.  FWD: sums of (y - pivot), (y - pivot)^2, the pivot, the count
    float ps1[8], ps2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { csum[e] = 0.f; ssq[e] = 0.f; spv[e] = 0.f; ps1[e] = 0.f; ps2[e] = 0.f; }

    const int64_t ntiles = (p.M + WROWS - 1) / WROWS;
    const int64_t G = gridDim.x;
    // stores through buffer descriptors: a row past the end gets an offset past the end and is dropped (no branch around a store)
    constexpr unsigned OOB = 0xfffffff0u;
    const unsigned nbytes = (unsigned)((size_t)p.M * C * sizeof(bf16_t));
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mid = __builtin_amdgcn_make_buffer_rsrc(MODE == 1 ? p.mid : p.out, 0, (int)nbytes, 0x00020000);

    v4f px[PIECES], py[MODE == 0 ? PIECES : 1], pg[MODE == 1 ? PIECES : 1];
    auto prefetch = [&](int64_t t, v4f (&px)[PIECES]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int64_t m = t * WROWS + prow + RSTEP * i;
            const size_t off = (size_t)(m < p.M ? m : p.M - 1) * C + cc8;     // clamped: always inside the tensor
            px[i] = *reinterpret_cast<const v4f *>(p.x + off);
            if (MODE == 1) pg[i] = *reinterpret_cast<const v4f *>(p.g + off);
        }
    };
    auto stage = [&](int64_t t, bf16_t *tile, v4f (&px)[PIECES]) __attribute__((always_inline)) {     // registers -> transform -> LDS (backward: dh -> global too)
        float k0[8], k1[8], k2[8];
        consts(0, k0); consts(1, k1);
        if (MODE == 1) consts(2, k2);
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int row = prow + RSTEP * i;
            const int64_t m = t * WROWS + row;
            float xv[8], o[8];
            Elem<bf16_t>::unpack16(px[i], xv);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaxf(__builtin_fmaf(xv[e], k0[e], k1[e]), 0.f);
            } else {
                float gv[8];
                Elem<bf16_t>::unpack16(pg[i], gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(k0[e], gv[e], -__builtin_fmaf(k1[e], xv[e], k2[e]));
            }
            const v4u pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
            *reinterpret_cast<v4u *>(tile + row * PITCH + cc8) = pk;
            if (MODE == 1) {
                const bool in = m < p.M;
                __builtin_amdgcn_raw_buffer_store_b128(u4{pk.x, pk.y, pk.z, pk.w}, rs_mid, (int)(in ? (unsigned)(((size_t)m * C + cc8) * 2) : OOB), 0, 0);
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[e] += in ? o[e] : 0.f;
            }
        }
    };

    prefetch(blockIdx.x, px);
    stage(blockIdx.x, tin, px);
    if constexpr (MODE == 0) {
        prefetch(blockIdx.x + G, px);
        prefetch(blockIdx.x + 2 * G, py);
    }
    __syncthreads();
    // pn: the register set the NEXT tile is staged from (FWD: it already holds tile t + G; BWD: loaded here)
    auto one_tile = [&](int64_t t, int cur, v4f (&pn)[PIECES]) __attribute__((always_inline)) {
        const int64_t nxt = t + G;
        if (MODE == 1) prefetch(nxt, pn);
        v4f ph[MODE == 1 ? PIECES : 1];
        if (prev) {                            // the rows of prev_x under THIS tile: in flight across the MFMA loop, used in the store phase
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int64_t m = t * WROWS + prow + RSTEP * i;
                ph[i] = *reinterpret_cast<const v4f *>(p.prev_x + (size_t)(m < p.M ? m : p.M - 1) * C + cc8);
            }
        }
        const bf16_t *tile = tin + cur * WROWS * PITCH;
        v16f acc[WROWS / 32];
#pragma unroll
        for (int rb = 0; rb < WROWS / 32; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
        const bf16_t *brow = tile + l31 * PITCH + 8 * hb;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int rb = 0; rb < WROWS / 32; ++rb) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4f *>(brow + 32 * rb * PITCH + 16 * ks));
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[ks], b, acc[rb], 0, 0, 0);
            }
        // lane = row (32 rb + l31); registers 4q .. 4q+3 = channels 32 wave + 8 q + 4 hb .. +3
#pragma unroll
        for (int rb = 0; rb < WROWS / 32; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = 32 * wave + 8 * q + 4 * hb;
                const v4f bv = *reinterpret_cast<const v4f *>(sbias + ch);
                const v2u pk = {pack_bf16(acc[rb][4 * q] + bv.x, acc[rb][4 * q + 1] + bv.y), pack_bf16(acc[rb][4 * q + 2] + bv.z, acc[rb][4 * q + 3] + bv.w)};
                *reinterpret_cast<v2u *>(tout + (32 * rb + l31) * PITCH + ch) = pk;
            }
        __syncthreads();                       // the output tile is complete (and every wave is done reading this input tile)
        float pmu[8], pfs[8], pbe[8];
        if (prev) { consts(3, pmu); consts(4, pfs); consts(5, pbe); }
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int row = prow + RSTEP * i;
            const int64_t m = t * WROWS + row;
            const bool in = m < p.M;
            const v4f piece = *reinterpret_cast<const v4f *>(tout + row * PITCH + cc8);
            const u4 pu = __builtin_bit_cast(u4, piece);
            __builtin_amdgcn_raw_buffer_store_b128(pu, rs_out, (int)(in ? (unsigned)(((size_t)m * C + cc8) * 2) : OOB), 0, 0);
            if (prev) {                    // bn_bwd_partial_kernel's sums on (prev_x, the dx values as stored), same mask expression
                float dv[8], hv[8];
                Elem<bf16_t>::unpack16(piece, dv);
                Elem<bf16_t>::unpack16(ph[i], hv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float hc = hv[e] - pmu[e];
                    const float ge = (in && (hc * pfs[e] + pbe[e]) > 0.f) ? dv[e] : 0.f;
                    ps1[e] += ge;
                    ps2[e] = __builtin_fmaf(ge, hc, ps2[e]);
                }
            }
            if (MODE == 0 && p.stat_tiles && in) {          // one pass about a pivot (the thread's first value), of the values as stored
                float yv[8];
                Elem<bf16_t>::unpack16(piece, yv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    spv[e] = scnt == 0.f ? yv[e] : spv[e];
                    const float d = yv[e] - spv[e];
                    csum[e] += d;
                    ssq[e] = __builtin_fmaf(d, d, ssq[e]);
                }
                scnt += 1.f;
            }
        }
        stage(nxt, tin + (cur ^ 1) * WROWS * PITCH, pn);       // (a tile past the end: clamped loads, dropped stores, zero sums)
        if (MODE == 0) prefetch(t + 3 * G, pn);
        __syncthreads();                       // the output tile has left LDS; the next input tile is in place
    };
    if constexpr (MODE == 0) {
        for (int64_t t = blockIdx.x; t < ntiles; t += 2 * G) {
            one_tile(t, 0, px);
            if (t + G < ntiles) one_tile(t + G, 1, py);
        }
    } else {
        int cur = 0;
        for (int64_t t = blockIdx.x; t < ntiles; t += G, cur ^= 1) one_tile(t, cur, px);
    }

    // ---- closing reductions: threads with the same channel group (tid % CPR) are combined in row-slot order ----
    float *red = reinterpret_cast<float *>(smem_raw);      // [17][NT] floats over the tiles (the loop ended on a barrier)
    if (MODE == 0 && p.stat_tiles) {
        const float inv = scnt > 0.f ? 1.f / scnt : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[e * NT + tid] = spv[e] + csum[e] * inv;
            red[(8 + e) * NT + tid] = fmaxf(ssq[e] - csum[e] * csum[e] * inv, 0.f);
        }
        red[16 * NT + tid] = scnt;
        __syncthreads();
        for (int c = tid; c < C; c += NT) {
            const int grp = c >> 3, e = c & 7;
            double N = 0.0, S = 0.0;
            for (int r = 0; r < RSTEP; ++r) {
                const double n = red[16 * NT + r * CPR + grp];
                N += n;
                S += n * (double)red[e * NT + r * CPR + grp];
            }
            const double mu = N > 0.0 ? S / N : 0.0;
            double Q = 0.0;
            for (int r = 0; r < RSTEP; ++r) {
                const double n = red[16 * NT + r * CPR + grp];
                const double dl = (double)red[e * NT + r * CPR + grp] - mu;
                Q += n > 0.0 ? (double)red[(8 + e) * NT + r * CPR + grp] + n * dl * dl : 0.0;
            }
            float *dst = p.stat_tiles + (size_t)blockIdx.x * 3 * C;
            dst[c] = (float)N;
            dst[C + c] = (float)S;
            dst[2 * C + c] = (float)Q;
        }
    }
    if (prev) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[e * NT + tid] = ps1[e]; red[(8 + e) * NT + tid] = ps2[e]; }
        __syncthreads();
        for (int c = tid; c < C; c += NT) {
            const int grp = c >> 3, e = c & 7;
            float a = 0.f, b = 0.f;
            for (int r = 0; r < RSTEP; ++r) { a += red[e * NT + r * CPR + grp]; b += red[(8 + e) * NT + r * CPR + grp]; }
            p.prev_partial[(size_t)blockIdx.x * 2 * C + c] = a;
            p.prev_partial[(size_t)blockIdx.x * 2 * C + C + c] = b * p.prev_invstd[c];
        }
        __syncthreads();
    }
    if (MODE == 1 && p.colsum_partial) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[e * NT + tid] = csum[e];
        __syncthreads();
        for (int c = tid; c < C; c += NT) {
            const int grp = c >> 3, e = c & 7;
            float sacc = 0.f;
            for (int r = 0; r < RSTEP; ++r) sacc += red[e * NT + r * CPR + grp];
            p.colsum_partial[(size_t)blockIdx.x * C + c] = sacc;
        }
    }
}

template <int MODE>
int launch_flat_wide(const FlatParams &p, int blocks, hipStream_t s)
{
    constexpr int C = 256;
    const size_t lds = (size_t)3 * WROWS * (C + 8) * sizeof(bf16_t);      // >= the [17][512] floats of the closing reductions
    static_assert((size_t)3 * WROWS * (C + 8) * sizeof(bf16_t) >= (size_t)17 * 512 * sizeof(float), "closing reductions fit in the tiles");
    static LdsOptIn once;
    const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&flat_wide_kernel<MODE>)}, lds, "flat_gemm (C = 256)");
    if (rc != NSG_OK) return rc;
    hipLaunchKernelGGL((flat_wide_kernel<MODE>), dim3(blocks), dim3(512), lds, s, p);
    return nsg_check_launch("flat_wide_kernel");
}

// ------------------------------------------------------------------------------------------------
// C = 128: the backward of the 1x1 conv in ONE pass -- flat_gemm_kernel<4, 1> (dh = BatchNorm backward of dy, dx = dh * W, the
// sums of the BatchNorm + ReLU in front) AND the conv's weight gradient dw[co][ci] = sum_rows dh[row][co] * a[row][ci] with
// a = relu(bn_front(prev_x)) rebuilt from the prev_x rows the kernel loads anyway.  dh is never stored (the weight gradient was
// its only reader) and prev_x / dh are not read a second time by a weight-gradient kernel: 4 tensor passes instead of 7.
// 512 threads, one block per CU.  Wave w: data gradient rows 32 (w & 3) .. +31 x input channels 64 (w >> 2) .. +63;
// weight gradient output channels 32 (w & 3) .. +31 x input channels 64 (w >> 2) .. +63, accumulated over all the block's tiles
// (two MFMA tiles in registers), operands read TRANSPOSED from the row-major LDS tiles (ds_read_b64_tr_b16, as wgrad_gemm_bf16).
// Per block a [128][128] fp32 partial of dw; nsg_launch_wgrad_reduce adds them in block order.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ s16x4 flat_tr_read(const bf16_t *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
}

__global__ __launch_bounds__(512) void flat_bwd_fused_kernel(const FlatParams p, float *__restrict__ dw_partial)
{
    constexpr int C = 128, NT = 512, KS = C / 16, PITCH = C + 8, CPR = C / 8, RSTEP = NT / CPR, PIECES = ROWS / RSTEP;   // 32 rows per step, 4 pieces
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t *wt = reinterpret_cast<bf16_t *>(smem_raw);            // [C (n = ci)][PITCH (k = co)]
    bf16_t *tdh = wt + C * PITCH;                                 // [ROWS][PITCH] dh
    bf16_t *ta = tdh + ROWS * PITCH;                              // [ROWS][PITCH] a = relu(bn_front(prev_x))
    bf16_t *tout = ta + ROWS * PITCH;                             // [ROWS][PITCH] dx on its way out
    __shared__ __attribute__((aligned(16))) float sconst[8 * C];  // sc, k1, k2 | front: mean, fs, beta | front: fs, off (the forward's form)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hb = lane >> 5;
    const int wr = wave & 3, wc = wave >> 2;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));

    {   // W^T -> LDS as bf16 [n = ci][k = co]: lanes along n (rows k, k + 1 of w), all loads first
        constexpr int WIT = C * C / 2 / NT;
        float wa[WIT], wb[WIT];
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + NT * it, n = i % C, k = (i / C) * 2;
            wa[it] = p.w[(size_t)k * C + n];
            wb[it] = p.w[(size_t)(k + 1) * C + n];
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + NT * it, n = i % C, k = (i / C) * 2;
            *reinterpret_cast<unsigned *>(wt + n * PITCH + k) = pack_bf16(wa[it], wb[it]);
        }
    }
    if (tid < C) {
        const int c = tid;
        const float sc = p.gamma[c] * p.invstd[c];
        const float k1 = sc * p.invstd[c] * (p.dgamma[c] * p.inv_m);
        sconst[c] = sc;
        sconst[C + c] = k1;
        sconst[2 * C + c] = __builtin_fmaf(sc, p.dbeta[c] * p.inv_m, -(k1 * p.mean[c]));
        const float fs = p.prev_invstd[c] * p.prev_gamma[c];
        sconst[3 * C + c] = p.prev_mean[c];
        sconst[4 * C + c] = fs;
        sconst[5 * C + c] = p.prev_beta[c];
        sconst[6 * C + c] = fs;
        sconst[7 * C + c] = __builtin_fmaf(-p.prev_mean[c], fs, p.prev_beta[c]);     // the forward's staging constants (flat_gemm_kernel MODE 0)
    }
    __syncthreads();
    const int prow = tid / CPR, cc8 = (tid % CPR) * 8;
    auto consts = [&](int j, float (&v)[8]) __attribute__((always_inline)) {
        const v4f a = *reinterpret_cast<const v4f *>(sconst + j * C + cc8), b = *reinterpret_cast<const v4f *>(sconst + j * C + cc8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    };
    float csum[8], ps1[8], ps2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { csum[e] = 0.f; ps1[e] = 0.f; ps2[e] = 0.f; }
    const int64_t ntiles = (p.M + ROWS - 1) / ROWS;
    const int64_t G = gridDim.x;
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(unsigned)((size_t)p.M * C * sizeof(bf16_t)), 0x00020000);

    // register sets: (h, dy, prev_x) of the tile being staged; prev_x of the tile in LDS is kept for its store phase
    v4f px[PIECES], pg[PIECES], ph[PIECES], phc[PIECES];
    auto prefetch = [&](int64_t t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int64_t m = t * ROWS + prow + RSTEP * i;
            const size_t off = (size_t)(m < p.M ? m : p.M - 1) * C + cc8;     // clamped: always inside the tensors
            px[i] = *reinterpret_cast<const v4f *>(p.x + off);
            pg[i] = *reinterpret_cast<const v4f *>(p.g + off);
            ph[i] = *reinterpret_cast<const v4f *>(p.prev_x + off);
        }
    };
    auto stage = [&](int64_t t) __attribute__((always_inline)) {       // registers -> dh and a -> LDS; rows past the end are ZERO in both (they enter dw)
        float k0[8], k1[8], k2[8], fs[8], fo[8];
        consts(0, k0); consts(1, k1); consts(2, k2); consts(6, fs); consts(7, fo);
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int row = prow + RSTEP * i;
            const bool in = t * ROWS + row < p.M;
            float xv[8], gv[8], hv[8], o[8], a[8];
            Elem<bf16_t>::unpack16(px[i], xv);
            Elem<bf16_t>::unpack16(pg[i], gv);
            Elem<bf16_t>::unpack16(ph[i], hv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = in ? __builtin_fmaf(k0[e], gv[e], -__builtin_fmaf(k1[e], xv[e], k2[e])) : 0.f;
                a[e] = in ? fmaxf(__builtin_fmaf(hv[e], fs[e], fo[e]), 0.f) : 0.f;
                csum[e] += o[e];
            }
            const v4u pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
            const v4u pa = {pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(a[4], a[5]), pack_bf16(a[6], a[7])};
            *reinterpret_cast<v4u *>(tdh + row * PITCH + cc8) = pk;
            *reinterpret_cast<v4u *>(ta + row * PITCH + cc8) = pa;
            phc[i] = ph[i];
        }
    };
    v16f dw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[j][r] = 0.f;

    prefetch(blockIdx.x);
    stage(blockIdx.x);
    __syncthreads();
    // transposing reads: lane = 32 h + 16 g16 + 4 q + pp supplies row 8 h + q (+ 4), columns 16 g16 + 4 pp .. + 3 of a 32-channel block
    const int g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
    const bf16_t *trA = tdh + (8 * hb + q) * PITCH + 32 * wr + 16 * g16 + 4 * pp;
    const bf16_t *trB = ta + (8 * hb + q) * PITCH + 64 * wc + 16 * g16 + 4 * pp;
    for (int64_t t = blockIdx.x; t < ntiles; t += G) {
        const int64_t nxt = t + G;
        prefetch(nxt);                         // (past the end: clamped loads, staged as zeros)
        // ---- data gradient: out^T[n = ci][row] = W^T[ci][co] * dh^T[co][row] ----
        v16f acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const bf16_t *brow = tdh + (32 * wr + l31) * PITCH + 8 * hb;
        const bf16_t *arow = wt + (64 * wc + l31) * PITCH + 8 * hb;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4f *>(brow + 16 * ks));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4f *>(arow + 32 * j * PITCH + 16 * ks));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
            }
        }
        // ---- weight gradient: dw[co][ci] += sum over the tile's rows of dh[row][co] * a[row][ci] ----
#pragma unroll
        for (int ks = 0; ks < ROWS / 16; ++ks) {
            const s16x4 alo = flat_tr_read(trA + (16 * ks) * PITCH), ahi = flat_tr_read(trA + (16 * ks + 4) * PITCH);
            const s16x8 a = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const s16x4 blo = flat_tr_read(trB + (16 * ks) * PITCH + 32 * j), bhi = flat_tr_read(trB + (16 * ks + 4) * PITCH + 32 * j);
                const s16x8 b = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
                dw[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), dw[j], 0, 0, 0);
            }
        }
        // dx block -> LDS: lane = row 32 wr + l31; registers 4 qq .. 4 qq + 3 of block j = channels 64 wc + 32 j + 8 qq + 4 hb .. + 3
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int ch = 64 * wc + 32 * j + 8 * qq + 4 * hb;
                const v2u pk = {pack_bf16(acc[j][4 * qq], acc[j][4 * qq + 1]), pack_bf16(acc[j][4 * qq + 2], acc[j][4 * qq + 3])};
                *reinterpret_cast<v2u *>(tout + (32 * wr + l31) * PITCH + ch) = pk;
            }
        __syncthreads();                       // dx is complete; every wave is done with dh and a
        float pmu[8], pfs[8], pbe[8];
        consts(3, pmu); consts(4, pfs); consts(5, pbe);
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int row = prow + RSTEP * i;
            const int64_t m = t * ROWS + row;
            const bool in = m < p.M;
            const v4f piece = *reinterpret_cast<const v4f *>(tout + row * PITCH + cc8);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, piece), rs_out, (int)(in ? (unsigned)(((size_t)m * C + cc8) * 2) : OOB), 0, 0);
            float dv[8], hv[8];                // bn_bwd_partial_kernel's sums on (prev_x, the dx values as stored), same mask expression
            Elem<bf16_t>::unpack16(piece, dv);
            Elem<bf16_t>::unpack16(phc[i], hv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float hc = hv[e] - pmu[e];
                const float ge = (in && (hc * pfs[e] + pbe[e]) > 0.f) ? dv[e] : 0.f;
                ps1[e] += ge;
                ps2[e] = __builtin_fmaf(ge, hc, ps2[e]);
            }
        }
        stage(nxt);
        __syncthreads();
    }
    // ---- this block's share of dw: [co][ci] fp32; lane = ci column, registers = co rows (r & 3) + 8 (r >> 2) + 4 hb ----
    {
        float *dst = dw_partial + (size_t)blockIdx.x * C * C;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * hb;
                dst[(size_t)co * C + 64 * wc + 32 * j + l31] = dw[j][r];
            }
    }
    // ---- closing reductions over the threads of a channel group, in row-slot order ----
    float *red = reinterpret_cast<float *>(smem_raw);      // [16][NT] floats (the loop ended on a barrier)
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[e * NT + tid] = ps1[e]; red[(8 + e) * NT + tid] = ps2[e]; }
    __syncthreads();
    for (int c = tid; c < C; c += NT) {
        const int grp = c >> 3, e = c & 7;
        float a = 0.f, b = 0.f;
        for (int r = 0; r < RSTEP; ++r) { a += red[e * NT + r * CPR + grp]; b += red[(8 + e) * NT + r * CPR + grp]; }
        p.prev_partial[(size_t)blockIdx.x * 2 * C + c] = a;
        p.prev_partial[(size_t)blockIdx.x * 2 * C + C + c] = b * p.prev_invstd[c];
    }
    __syncthreads();
    if (p.colsum_partial) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[e * NT + tid] = csum[e];
        __syncthreads();
        for (int c = tid; c < C; c += NT) {
            const int grp = c >> 3, e = c & 7;
            float sacc = 0.f;
            for (int r = 0; r < RSTEP; ++r) sacc += red[e * NT + r * CPR + grp];
            p.colsum_partial[(size_t)blockIdx.x * C + c] = sacc;
        }
    }
}

template <int NB, int MODE>
int launch_flat(const FlatParams &p, int blocks, hipStream_t s)
{
    constexpr int C = 32 * NB;
    size_t lds = (size_t)(ROWS + C) * (C + 8) * sizeof(bf16_t);
    if (lds < 17 * 256 * sizeof(float)) lds = 17 * 256 * sizeof(float);      // the closing reductions reuse the front of it as [17][256] floats
    static LdsOptIn once;
    if (lds > 65536 - 1024) {
        const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&flat_gemm_kernel<NB, MODE>)}, lds, "flat_gemm");
        if (rc != NSG_OK) return rc;
    }
    hipLaunchKernelGGL((flat_gemm_kernel<NB, MODE>), dim3(blocks), dim3(256), lds, s, p);
    return nsg_check_launch("flat_gemm_kernel");
}

template <int MODE>
int dispatch_flat(const FlatParams &p, int C, int blocks, hipStream_t s)
{
    if (C == 256) return launch_flat_wide<MODE>(p, blocks, s);
    switch (C / 32) {
    case 1: return launch_flat<1, MODE>(p, blocks, s);
    case 2: return launch_flat<2, MODE>(p, blocks, s);
    default: return launch_flat<4, MODE>(p, blocks, s);
    }
}

constexpr int FLAT_WIDE_BLOCKS = 256;  // C = 256: 1 block of 8 waves per CU
constexpr int FLAT_BLOCKS = 512;      // 2 blocks per CU are resident (LDS): one prologue per block slot

int flat_blocks(int64_t M, int C)
{
    const int rows = C == 256 ? WROWS : ROWS, cap = C == 256 ? FLAT_WIDE_BLOCKS : FLAT_BLOCKS;
    const int64_t nt = (M + rows - 1) / rows;
    return (int)(nt < cap ? nt : cap);
}

}  // namespace

// (the staging keeps a thread on one channel group: 256 threads must be a whole number of tile rows, i.e. C / 8 divides 256)
bool nsg_flat1x1_supported(int dtype, int C) { return dtype == NSG_BF16 && (C == 32 || C == 64 || C == 128 || C == 256); }

size_t nsg_flat1x1_workspace_bytes(int C)
{
    const size_t bwd = nsg_align_up((size_t)FLAT_BLOCKS * C * sizeof(float), 256) + nsg_align_up((size_t)FLAT_BLOCKS * 2 * C * sizeof(float), 256);
    const size_t fwd = nsg_align_up(nsg_bn_tiles_bytes(FLAT_BLOCKS, C), 256);
    return bwd > fwd ? bwd : fwd;
}

// y = relu((x - mean) * invstd * gamma + beta) * W^T + bias
// stat_tiles_out != null: the workspace receives one (count, sum, M2) record per block for nsg_bn_stats_from_tiles; *nblocks = their number
int nsg_launch_flat1x1_forward(const void *x, const float *mean, const float *invstd, const float *gamma, const float *beta, const float *w,
                               const float *bias, void *y, int64_t M, int C, void *ws, int want_stats, int *nblocks, hipStream_t s)
{
    FlatParams p = {};
    p.x = reinterpret_cast<const bf16_t *>(x); p.w = w; p.bias = bias; p.out = reinterpret_cast<bf16_t *>(y); p.M = M;
    p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta;
    p.stat_tiles = want_stats ? reinterpret_cast<float *>(ws) : nullptr;
    const int blocks = flat_blocks(M, C);
    if (nblocks) *nblocks = blocks;
    if (C == 256 && !nsg_aligned16(w)) return nsg_fail(NSG_E_INVALID, "flat_gemm (C = 256): w must be 16-byte aligned");
    return dispatch_flat<0>(p, C, blocks, s);
}

// The backward of the 1x1 conv in one pass (C = 128): dx, the column sums of dh, the sums of the BatchNorm in front, and dw
// (partials per block in the workspace: *dw_partial [blocks][C][C], summed by the caller).  dh itself is not stored.
constexpr int FLAT_FUSED_BLOCKS = 256;
bool nsg_flat1x1_fused_bwd_supported(int dtype, int C) { return dtype == NSG_BF16 && C == 128; }
size_t nsg_flat1x1_fused_bwd_workspace_bytes(int C)
{
    return nsg_align_up((size_t)FLAT_FUSED_BLOCKS * C * sizeof(float), 256) + nsg_align_up((size_t)FLAT_FUSED_BLOCKS * 2 * C * sizeof(float), 256) +
           (size_t)FLAT_FUSED_BLOCKS * C * C * sizeof(float);
}
int nsg_launch_flat1x1_fused_bwd(const void *h, const void *dy, const float *mean, const float *invstd, const float *gamma, const float *dgamma,
                                 const float *dbeta, const float *w, void *dx, float *dw, int64_t M, int C, void *ws, int *nblocks,
                                 const void *prev_x, const float *prev_mean, const float *prev_invstd, const float *prev_gamma,
                                 const float *prev_beta, float **colsum_partial, float **prev_partial, float **dw_partial, hipStream_t s)
{
    (void)dw;       // (the caller sums the block partials of dw together with the BatchNorm sums: nsg_launch_bn_bwd_final_wreduce)
    char *wsb = reinterpret_cast<char *>(ws);
    FlatParams p = {};
    p.x = reinterpret_cast<const bf16_t *>(h); p.g = reinterpret_cast<const bf16_t *>(dy); p.w = w;
    p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.dgamma = dgamma; p.dbeta = dbeta; p.inv_m = 1.f / (float)M;
    p.prev_x = reinterpret_cast<const bf16_t *>(prev_x);
    p.prev_mean = prev_mean; p.prev_invstd = prev_invstd; p.prev_gamma = prev_gamma; p.prev_beta = prev_beta;
    p.colsum_partial = reinterpret_cast<float *>(wsb);
    p.prev_partial = reinterpret_cast<float *>(wsb + nsg_align_up((size_t)FLAT_FUSED_BLOCKS * C * sizeof(float), 256));
    float *dwp = reinterpret_cast<float *>(wsb + nsg_align_up((size_t)FLAT_FUSED_BLOCKS * C * sizeof(float), 256) +
                                           nsg_align_up((size_t)FLAT_FUSED_BLOCKS * 2 * C * sizeof(float), 256));
    p.out = reinterpret_cast<bf16_t *>(dx); p.M = M;
    const int64_t nt = (M + ROWS - 1) / ROWS;
    const int blocks = (int)(nt < FLAT_FUSED_BLOCKS ? nt : FLAT_FUSED_BLOCKS);
    *nblocks = blocks;
    *colsum_partial = p.colsum_partial;
    *prev_partial = p.prev_partial;
    const size_t lds = (size_t)(C + 3 * ROWS) * (C + 8) * sizeof(bf16_t);
    static LdsOptIn once;
    int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&flat_bwd_fused_kernel)}, lds, "flat_gemm (fused backward)");
    if (rc != NSG_OK) return rc;
    hipLaunchKernelGGL(flat_bwd_fused_kernel, dim3(blocks), dim3(512), lds, s, p, dwp);
    *dw_partial = dwp;
    return nsg_check_launch("flat_bwd_fused_kernel");
}

// dh = BatchNorm backward of dy at input h (no ReLU), stored; dx = dh * W; partial column sums of dh -> colsum_partial [blocks][C];
// returns the number of blocks through *nblocks
// prev_x != null: also the partial sums of the BatchNorm + ReLU in front of the conv -> [blocks][2][C] behind the column-sum partials
int nsg_launch_flat1x1_backward(const void *h, const void *dy, const float *mean, const float *invstd, const float *gamma, const float *dgamma,
                                const float *dbeta, const float *w, void *dh, void *dx, int64_t M, int C, void *ws, int *nblocks,
                                const void *prev_x, const float *prev_mean, const float *prev_invstd, const float *prev_gamma,
                                const float *prev_beta, float **prev_partial, hipStream_t s)
{
    float *partial = reinterpret_cast<float *>(ws);
    FlatParams p = {};
    p.x = reinterpret_cast<const bf16_t *>(h); p.g = reinterpret_cast<const bf16_t *>(dy); p.w = w;
    p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.dgamma = dgamma; p.dbeta = dbeta; p.inv_m = 1.f / (float)M;
    if (prev_x) {
        p.prev_x = reinterpret_cast<const bf16_t *>(prev_x);
        p.prev_mean = prev_mean; p.prev_invstd = prev_invstd; p.prev_gamma = prev_gamma; p.prev_beta = prev_beta;
        p.prev_partial = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + nsg_align_up((size_t)FLAT_BLOCKS * C * sizeof(float), 256));
        if (prev_partial) *prev_partial = p.prev_partial;
    }
    p.out = reinterpret_cast<bf16_t *>(dx); p.mid = reinterpret_cast<bf16_t *>(dh); p.colsum_partial = partial; p.M = M;
    const int blocks = flat_blocks(M, C);
    *nblocks = blocks;
    return dispatch_flat<1>(p, C, blocks, s);
}
