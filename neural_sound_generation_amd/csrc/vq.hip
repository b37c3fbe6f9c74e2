// Vector quantiser kernels (gfx950): fused ||x-e||^2 nearest-code search with a bit-exact argmin,
// code gather, codebook scatter-add, EMA update.   Reference: src/vector_quantization.py:6-66,
// src/models.py:132-142.
//
// Bit-exactness (DESIGN.md "bit-exact argmin"): the reference's CPU result is, entry by entry,
//     dist[i][k] = fl( fl(c2[k] + x2[i]) - 2*dot[i][k] ),   dot = fmaf chain over d = 0..D-1 from +0,
// with c2/x2 summed in ATen's vector-lane order, and torch.min keeps the first minimal index.
// v_mfma_f32_32x32x2_f32 IS a k-ordered fmaf chain (one rounding per product, no wider
// accumulator), so feeding it d = 2s (lanes 0-31) and d = 2s+1 (lanes 32-63) at step s reproduces
// the chain exactly on the matrix pipe.  tests/test_gpu_ops.py checks MFMA == VALU chain bit for bit
// (nsg_debug_dot) and the indices against fixtures generated from the reference.
#include "nsg_common.h"
#include <math.h>

#ifndef NSG_VQ_W1_DP
#define NSG_VQ_W1_DP 256      // from this padded width on: one wave per SIMD (no spills) instead of two
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// torch.sum(v**2, dim=1) in ATen's CPU order: 8 vector lanes x 4 interleaved accumulators.
// 8 threads per row: thread l plays vector lane l.
// ------------------------------------------------------------------------------------------------
__global__ void rowsumsq_kernel(const float *__restrict__ v, int64_t rows, int D, float *__restrict__ out)
{
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t row = gid >> 3;
    const int l = (int)(gid & 7);
    const bool live = row < rows;
    const float *p = v + (live ? row : 0) * (int64_t)D;
    const int nvec = D >> 3, ngrp = nvec >> 2;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int c = 0; c < ngrp; ++c) {
        const float v0 = p[(4 * c + 0) * 8 + l], v1 = p[(4 * c + 1) * 8 + l];
        const float v2 = p[(4 * c + 2) * 8 + l], v3 = p[(4 * c + 3) * 8 + l];
        a0 = __fadd_rn(a0, __fmul_rn(v0, v0));
        a1 = __fadd_rn(a1, __fmul_rn(v1, v1));
        a2 = __fadd_rn(a2, __fmul_rn(v2, v2));
        a3 = __fadd_rn(a3, __fmul_rn(v3, v3));
    }
    for (int i = ngrp * 4; i < nvec; ++i) {
        const float v0 = p[i * 8 + l];
        a0 = __fadd_rn(a0, __fmul_rn(v0, v0));
    }
    const float a = __fadd_rn(__fadd_rn(__fadd_rn(a0, a1), a2), a3);
    float fin = 0.f;
    for (int k = nvec * 8; k < D; ++k) fin = __fadd_rn(fin, __fmul_rn(p[k], p[k]));
    const int base = threadIdx.x & ~7;
#pragma unroll
    for (int j = 0; j < 8; ++j) fin = __fadd_rn(fin, __shfl(a, (base + j) & 63, 64));
    if (live && l == 0) out[row] = fin;
}

// ------------------------------------------------------------------------------------------------
// Fused search.  Block = 4 waves = 128 rows (32 per wave); codes streamed in tiles of 32 through
// double-buffered LDS; each wave keeps its 32 rows of x in DP/2 registers as MFMA A fragments.
// ------------------------------------------------------------------------------------------------
template <int DP, bool USE_MFMA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(USE_MFMA && DP < NSG_VQ_W1_DP ? 2 : 1))) void vq_forward_kernel(const float *__restrict__ x, const float *__restrict__ e,
                                                         const float *__restrict__ x2, const float *__restrict__ c2,
                                                         int64_t N, int D, int K, int64_t *__restrict__ idx_out,
                                                         float *__restrict__ codes_out, float *__restrict__ dmin_out,
                                                         int tiles_per_slice, float *__restrict__ part_d, int *__restrict__ part_i)
{
    constexpr int NS = DP / 2;           // MFMA steps (2 d's per step)
    // LDS image of a code row: [the NS even-d values | the NS odd-d values | 4 floats of padding].  Lane (l31, h) needs e[d = 2 s + h]
    // for s = 0 .. NS - 1 in order: in this image that is NS CONSECUTIVE floats from h * NS, fetched four steps per ds_read_b128;
    // pitch = DP + 4 floats = 4 (mod 64 banks): the 16 rows of a b128 lane group sit on 16 different 4-bank groups.  (Round 2 kept
    // the row as it is in memory and read one ds_read2_b32 per two steps; hipcc gave every read the same destination registers and
    // an s_waitcnt lgkmcnt(0), so each pair of MFMAs waited out a whole LDS round trip: 0.45-0.6 of the fp32 MFMA rate.)
    constexpr int EP = DP + 4;
    constexpr int XC = (DP < 64) ? DP : 64;  // x staging chunk (d's)
    constexpr int XP = XC + 1;
    static_assert(NS % 4 == 0, "four MFMA steps per 16-byte fragment read");
    constexpr int EJ = (32 * DP / 4 + 255) / 256;  // float4 per thread per code tile

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Es = smem;                                   // [2][32][EP]   (also x staging: [128][XP])
    __shared__ int sidx[128];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const bool vec_ok = (D & 3) == 0 && nsg_aligned16_dev(e);

    // ---- x rows -> A fragments: a[s] = x[row0 + wave*32 + l31][2s + h] ----
    float a[NS];
    if ((D & 3) == 0 && nsg_aligned16_dev(x)) {
        // 16-byte loads, all of a chunk's in flight at once, into an image de-interleaved like the code rows ([even d | odd d | pad],
        // pitch XC + 4): the lane's fragments are XC / 2 consecutive floats = XC / 8 ds_read_b128.  (Round 2 staged the rows with
        // one 4-byte load per thread per loop trip -- a serial chain of global-load latencies per block: 578 of the search's
        // 1 199 us at K = 512, D = 128 did not scale with K.)
        constexpr int XPV = XC + 4, Q4 = XC / 4, XJ = 128 * Q4 / 256;          // float4 per row chunk, per thread
        static_assert(128 * Q4 % 256 == 0 && XC % 8 == 0, "whole float4 per thread, whole b128 per fragment group");
        float *Xs = smem;  // [128][XPV]
        typedef float v2f_ __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int c = 0; c < DP / XC; ++c) {
            const int d0 = c * XC;
            v4f t[XJ];
#pragma unroll
            for (int i = 0; i < XJ; ++i) {
                const int f4 = tid + 256 * i;
                const int r = f4 / Q4, q = f4 - r * Q4;
                const int64_t row = row0 + r;
                const bool ok = row < N && d0 + 4 * q < D;
                const v4f v = *reinterpret_cast<const v4f *>(x + (ok ? row * D + d0 + 4 * q : 0));      // clamped, unconditional
                t[i] = ok ? v : v4f{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < XJ; ++i) {
                const int f4 = tid + 256 * i;
                const int r = f4 / Q4, q = f4 - r * Q4;
                *reinterpret_cast<v2f_ *>(Xs + r * XPV + 2 * q) = v2f_{t[i].x, t[i].z};
                *reinterpret_cast<v2f_ *>(Xs + r * XPV + XC / 2 + 2 * q) = v2f_{t[i].y, t[i].w};
            }
            __syncthreads();
            const float *src = Xs + (wave * 32 + l31) * XPV + h * (XC / 2);
#pragma unroll
            for (int j = 0; j < XC / 8; ++j) {
                const v4f v = *reinterpret_cast<const v4f *>(src + 4 * j);
                a[c * (XC / 2) + 4 * j] = v.x; a[c * (XC / 2) + 4 * j + 1] = v.y; a[c * (XC / 2) + 4 * j + 2] = v.z; a[c * (XC / 2) + 4 * j + 3] = v.w;
            }
            __syncthreads();
        }
    } else {
        float *Xs = smem;  // [128][XP]
#pragma unroll
        for (int c = 0; c < DP / XC; ++c) {
            const int d0 = c * XC;
            for (int f = tid; f < 128 * XC; f += 256) {
                const int r = f / XC, dd = f - r * XC;
                const int64_t row = row0 + r;
                const int d = d0 + dd;
                Xs[r * XP + dd] = (row < N && d < D) ? x[row * D + d] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < XC / 2; ++s) a[c * (XC / 2) + s] = Xs[(wave * 32 + l31) * XP + 2 * s + h];
            __syncthreads();
        }
    }

    // rows this lane's accumulator registers belong to, and their ||x||^2
    float x2v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        x2v[r] = row < N ? x2[row] : 0.f;
    }
    float best[16];
    int bidx[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { best[r] = INFINITY; bidx[r] = 0x7fffffff; }

    // this block's slice of the codebook: tiles [ct0, ct1) (the whole codebook unless the launch is split, nsg_vq_slices)
    const int ntiles_all = (K + 31) / 32;
    const int ct0 = (int)blockIdx.y * tiles_per_slice;
    const int ntiles = ct0 + tiles_per_slice < ntiles_all ? ct0 + tiles_per_slice : ntiles_all;
    v4f re[EJ];
    unsigned okmask = 0;
    // unconditional loads (clamped address), zero-fill at LDS-store time: no wait behind a load
    auto gload = [&](int ct) {
        unsigned mk = 0;
#pragma unroll
        for (int j = 0; j < EJ; ++j) {
            const int f = tid + 256 * j;       // float4 index inside the [32][DP] tile
            const int cr = f / (DP / 4), d4 = (f - cr * (DP / 4)) * 4;
            const int code = ct * 32 + cr;
            const bool row_ok = cr < 32 && code < K;
            if (vec_ok) {
                const bool ok = row_ok && d4 < D;       // D % 4 == 0: whole float4 inside the row
                re[j] = *reinterpret_cast<const v4f *>(e + (ok ? (size_t)code * D + d4 : 0));
                mk |= ok ? (1u << j) : 0u;
            } else {
                v4f v = {0.f, 0.f, 0.f, 0.f};
                if (row_ok) {
                    const float *src = e + (size_t)code * D + d4;
                    if (d4 + 0 < D) v.x = src[0];
                    if (d4 + 1 < D) v.y = src[1];
                    if (d4 + 2 < D) v.z = src[2];
                    if (d4 + 3 < D) v.w = src[3];
                }
                re[j] = v;
                mk |= 1u << j;
            }
        }
        okmask = mk;
    };
    auto lstore = [&](int buf) {
        float *es = Es + buf * 32 * EP;
#pragma unroll
        for (int j = 0; j < EJ; ++j) {
            const int f = tid + 256 * j;
            const int cr = f / (DP / 4), d4 = (f - cr * (DP / 4)) * 4;
            if (cr < 32) {
                const bool ok = (okmask >> j) & 1u;
                float *dst = es + cr * EP + (d4 >> 1);          // d4 % 4 == 0: (x, z) are the even d's, (y, w) the odd ones
                typedef float v2f_ __attribute__((ext_vector_type(2)));
                *reinterpret_cast<v2f_ *>(dst) = v2f_{ok ? re[j].x : 0.f, ok ? re[j].z : 0.f};
                *reinterpret_cast<v2f_ *>(dst + NS) = v2f_{ok ? re[j].y : 0.f, ok ? re[j].w : 0.f};
            }
        }
    };

    gload(ct0);
    lstore(0);
    __syncthreads();

    for (int ct = ct0; ct < ntiles; ++ct) {
        const int cur = (ct - ct0) & 1;
        if (ct + 1 < ntiles) gload(ct + 1);
        const int code = ct * 32 + l31;
        const float c2v = code < K ? c2[code] : INFINITY;
        const float *es = Es + cur * 32 * EP + l31 * EP + h * NS;

        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (USE_MFMA) {
            // the same chain, s = 0 .. NS - 1 in order; the fragment of steps 4 j + 4 .. + 7 is in flight under the MFMAs of 4 j .. + 3
            v4f bq[2];
            bq[0] = *reinterpret_cast<const v4f *>(es);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // (the first fragment: its own group, or the pairs below slip by one)
#pragma unroll
            for (int j = 0; j < NS / 4; ++j) {
                if (j + 1 < NS / 4) bq[(j + 1) & 1] = *reinterpret_cast<const v4f *>(es + 4 * (j + 1));
                const v4f b = bq[j & 1];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * j], b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * j + 1], b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * j + 2], b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * j + 3], b.w, acc, 0, 0, 0);
                if (j + 1 < NS / 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // the NEXT fragment's read, then this one's MFMAs
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        } else {
            // cross-check path: the same dot products as explicit fmaf chains on the vector ALU.
            // Lane (l31,h) needs x[row(r,h)][d] for every d: fetched from the lanes that hold it.
            const float *eb = Es + cur * 32 * EP + l31 * EP;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float e0 = eb[s], e1 = eb[NS + s];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int src_row = (r & 3) + 8 * (r >> 2) + 4 * h;  // row inside the wave's 32
                    const float xe = __shfl(a[s], src_row, 64);        // d = 2s   (held by lane src_row)
                    const float xo = __shfl(a[s], src_row + 32, 64);   // d = 2s+1 (held by lane src_row+32)
                    acc[r] = __fmaf_rn(xo, e1, __fmaf_rn(xe, e0, acc[r]));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float base = __fadd_rn(c2v, x2v[r]);
            const float dist = __fmaf_rn(-2.0f, acc[r], base);  // -2*dot is exact: one rounding, as addmm's epilogue
            if (dist < best[r]) { best[r] = dist; bidx[r] = code; }
        }
        if (ct + 1 < ntiles) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- first-minimum across the 32 lanes (codes) of each half; xor < 32 stays inside the half ----
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
        if (part_d != nullptr) {        // sliced launch: this slice's first minimum; nsg_launch_vq_combine finishes the job
            if (l31 == 0) {
                const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < N) { part_d[(size_t)blockIdx.y * N + row] = bv; part_i[(size_t)blockIdx.y * N + row] = bi; }
            }
            continue;
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
    if (codes_out == nullptr || part_d != nullptr) return;
    __syncthreads();
    // ---- gather: codes_out[row] = e[idx[row]]  (vector_quantization.py:40-42) ----
    if (vec_ok && nsg_aligned16_dev(codes_out)) {
        const int D4 = D >> 2;
        for (int f = tid; f < 128 * D4; f += 256) {
            const int r = f / D4, d4 = (f - r * D4) * 4;
            const int64_t row = row0 + r;
            if (row < N)
                *reinterpret_cast<v4f *>(codes_out + row * D + d4) = *reinterpret_cast<const v4f *>(e + (size_t)sidx[r] * D + d4);
        }
    } else {
        for (int f = tid; f < 128 * D; f += 256) {
            const int r = f / D, d = f - r * D;
            const int64_t row = row0 + r;
            if (row < N) codes_out[row * D + d] = e[(size_t)sidx[r] * D + d];
        }
    }
}

// dot products only, same fragment order as above (nsg_debug_dot)
template <bool USE_MFMA>
__global__ __launch_bounds__(64) void debug_dot_kernel(const float *x, const float *e, int N, int D, int K, float *out)
{
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int r0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    if (USE_MFMA) {
        v16f acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int s = 0; s < (D + 1) / 2; ++s) {
            const int d = 2 * s + h;
            const float av = d < D ? x[(size_t)(r0 + l31) * D + d] : 0.f;
            const float bv = d < D ? e[(size_t)(k0 + l31) * D + d] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            out[(size_t)row * K + k0 + l31] = acc[r];
        }
    } else {
        for (int rr = h; rr < 32; rr += 2) {
            float c = 0.f;
            for (int d = 0; d < D; ++d) c = __fmaf_rn(x[(size_t)(r0 + rr) * D + d], e[(size_t)(k0 + l31) * D + d], c);
            out[(size_t)(r0 + rr) * K + k0 + l31] = c;
        }
    }
}

// out[i][:] = e[idx[i]][:]
__global__ void gather_rows_kernel(const float *__restrict__ e, const int64_t *__restrict__ idx, int64_t N, int D, int K,
                                   float *__restrict__ out, int vec)
{
    if (vec) {
        const int D4 = D >> 2;
        const int64_t total = N * D4;
        for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < total; f += (int64_t)gridDim.x * blockDim.x) {
            const int64_t r = f / D4;
            const int d4 = (int)(f - r * D4) * 4;
            int64_t k = idx[r];
            k = k < 0 ? 0 : (k >= K ? K - 1 : k);
            *reinterpret_cast<v4f *>(out + r * D + d4) = *reinterpret_cast<const v4f *>(e + k * D + d4);
        }
    } else {
        const int64_t total = N * D;
        for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < total; f += (int64_t)gridDim.x * blockDim.x) {
            const int64_t r = f / D;
            const int d = (int)(f - r * D);
            int64_t k = idx[r];
            k = k < 0 ? 0 : (k >= K ? K - 1 : k);
            out[f] = e[k * D + d];
        }
    }
}

// Histogram of the indices: per-block bins in LDS (integer adds: order-free, exact), one global add per non-empty bin and
// block.  (Global atomics straight from every row serialise on the few hundred hot bins: 650 us for 655k rows.)
constexpr int COUNT_LDS_BINS = 8192;
__global__ __launch_bounds__(256) void count_codes_kernel(const int64_t *__restrict__ idx, int64_t N, int K, int *__restrict__ counts)
{
    extern __shared__ int bins[];
    const bool lds = K <= COUNT_LDS_BINS;
    if (lds) {
        for (int k = threadIdx.x; k < K; k += 256) bins[k] = 0;
        __syncthreads();
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = idx[i];
        if (k >= 0 && k < K) atomicAdd(lds ? &bins[k] : &counts[k], 1);
    }
    if (lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += 256)
            if (bins[k]) atomicAdd(&counts[k], bins[k]);
    }
}
__global__ void counts_to_float_kernel(const int *counts, int K, float *out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) out[k] = (float)counts[k];
}

// EMA codebook update (extension).  Kernel 1 (one block): decay the counts and form their total in a
// fixed order, so every rank / launch computes the identical value.  Kernel 2: the K x D part.
__global__ void ema_counts_kernel(float *ema_n, const float *n, int K, float decay, float *total)
{
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double part = 0.0;
    for (int k = tid; k < K; k += 256) {
        const float v = decay * ema_n[k] + (1.f - decay) * n[k];
        ema_n[k] = v;
        part += (double)v;
    }
    red[tid] = part;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < 256; ++i) t += red[i];
        total[0] = (float)t;
    }
}
__global__ void ema_codes_kernel(float *e, const float *ema_n, float *ema_s, const float *s, int K, int D, float decay,
                                 float eps, const float *total)
{
    const float tot = total[0];
    const int64_t n = (int64_t)K * D;
    for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < n; f += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(f / D);
        const float sv = decay * ema_s[f] + (1.f - decay) * s[f];
        ema_s[f] = sv;
        const float nn = (ema_n[k] + eps) / (tot + K * eps) * tot;
        e[f] = sv / nn;
    }
}

// First minimum over the S slices of a sliced search (slice order = code order, strict <: the first minimal index), then the
// gathers the unsplit kernels do themselves.  One block per 128 rows.
__global__ __launch_bounds__(256) void vq_combine_kernel(const float *__restrict__ pd, const int *__restrict__ pi, int S, int64_t N, int D, int K,
                                                         const float *__restrict__ e, int64_t *__restrict__ idx_out,
                                                         float *__restrict__ codes_out, float *__restrict__ dmin_out,
                                                         bf16_t *__restrict__ codes_lp, int lp_relu,
                                                         const float *__restrict__ clip_rows, int64_t rows_per_clip)
{
    __shared__ int sidx[128];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    if (tid < 128) {
        const int64_t row = row0 + tid;
        int bi = 0;
        if (row < N) {
            float bv = INFINITY;
            bi = 0x7fffffff;
            for (int sl = 0; sl < S; ++sl) {
                const float v = pd[(size_t)sl * N + row];
                const int i = pi[(size_t)sl * N + row];
                if (v < bv) { bv = v; bi = i; }
            }
            if (bi == 0x7fffffff) bi = 0;
            idx_out[row] = (int64_t)bi;
            if (dmin_out) dmin_out[row] = bv;
        }
        sidx[tid] = bi;
    }
    if (codes_out == nullptr && codes_lp == nullptr) return;
    __syncthreads();
    const bool vec = (D & 7) == 0 && nsg_aligned16_dev(e) && (!codes_out || nsg_aligned16_dev(codes_out));
    if (!vec) {         // (odd widths: the fp32 search only; the bf16 outputs need D % 8 == 0)
        for (int f = tid; f < 128 * D; f += 256) {
            const int r = f / D, d = f - r * D;
            if (row0 + r < N && codes_out) codes_out[(row0 + r) * D + d] = e[(size_t)sidx[r] * D + d];
        }
        return;
    }
    const int D8 = D >> 3;
    for (int f = tid; f < 128 * D8; f += 256) {
        const int r = f / D8, d8 = (f - r * D8) * 8;
        const int64_t row = row0 + r;
        if (row >= N) continue;
        const float *src = e + (size_t)sidx[r] * D + d8;
        const v4f a = *reinterpret_cast<const v4f *>(src), b = *reinterpret_cast<const v4f *>(src + 4);
        if (codes_out) {
            *reinterpret_cast<v4f *>(codes_out + row * D + d8) = a;
            *reinterpret_cast<v4f *>(codes_out + row * D + d8 + 4) = b;
        }
        if (codes_lp) {
            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            if (clip_rows) {
                const float *cr = clip_rows + (size_t)(row / rows_per_clip) * D + d8;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += cr[i];
            }
            if (lp_relu) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            Elem<bf16_t>::store16(codes_lp + row * D + d8, v);
        }
    }
}

template <int DP>
int launch_vq(bool mfma, const float *x, const float *e, const float *x2, const float *c2, int64_t N, int D, int K,
              int64_t *idx, float *codes, float *dmin, float *part_d, int *part_i, hipStream_t s)
{
    constexpr int XC = (DP < 64) ? DP : 64;
    const size_t lds_e = (size_t)2 * 32 * (DP + 4) * sizeof(float);
    const size_t lds_x = (size_t)128 * (XC + 4) * sizeof(float);
    const size_t lds = lds_e > lds_x ? lds_e : lds_x;
    const int64_t nb = nsg_cdiv(N, 128);
    if (nb > 0x7fffffff) return nsg_fail(NSG_E_UNSUPPORTED, "vq_forward: too many rows");
    static LdsOptIn once;
    if (lds > 65536 - 1024) {
        const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&vq_forward_kernel<DP, true>),
                                             reinterpret_cast<const void *>(&vq_forward_kernel<DP, false>)}, lds, "vq_forward");
        if (rc != NSG_OK) return rc;
    }
    const int S = part_d ? nsg_vq_slices(N, D, K) : 1;
    const int tps = (int)nsg_cdiv(nsg_cdiv(K, 32), S);
    if (mfma)
        hipLaunchKernelGGL((vq_forward_kernel<DP, true>), dim3((unsigned)nb, (unsigned)S), dim3(256), lds, s, x, e, x2, c2, N, D, K, idx, codes, dmin, tps, part_d, part_i);
    else
        hipLaunchKernelGGL((vq_forward_kernel<DP, false>), dim3((unsigned)nb, (unsigned)S), dim3(256), lds, s, x, e, x2, c2, N, D, K, idx, codes, dmin, tps, part_d, part_i);
    int rc = nsg_check_launch("vq_forward_kernel");
    if (rc == NSG_OK && part_d) rc = nsg_launch_vq_combine(part_d, part_i, S, N, D, K, e, idx, codes, dmin, nullptr, 0, nullptr, 0, s);
    return rc;
}

}  // namespace

int nsg_vq_slices(int64_t N, int D, int K)
{
    const int slots = NSG_SEARCH_CUS * (D > 128 ? 1 : 2);
    const int64_t nb = nsg_cdiv(N, 128);
    if (nb <= 0) return 1;
    int best = 1;
    double best_eff = 0.0;
    for (int S = 1; S <= 8; S *= 2) {
        if (S > 1 && nsg_cdiv(K, 32) / S < 16) break;                       // a slice keeps at least 16 code tiles (a block's fixed costs)
        const int64_t units = nb * S;
        const double eff = (double)units / (double)(nsg_cdiv(units, slots) * slots);
        if (eff > best_eff * 1.05) { best_eff = eff; best = S; }            // more slices only for a real gain (they cost a combine pass)
    }
    return best;
}

size_t nsg_vq_slice_bytes(int64_t N, int D, int K)
{
    const int S = nsg_vq_slices(N, D, K);
    return S > 1 ? 2 * nsg_align_up((size_t)S * (size_t)N * sizeof(float), 256) : 0;
}

int nsg_launch_vq_combine(const float *pd, const int *pi, int S, int64_t N, int D, int K, const float *e, int64_t *idx, float *codes,
                          float *dmin, bf16_t *codes_lp, int lp_relu, const float *clip_rows, int64_t rows_per_clip, hipStream_t s)
{
    hipLaunchKernelGGL(vq_combine_kernel, dim3((unsigned)nsg_cdiv(N, 128)), dim3(256), 0, s, pd, pi, S, N, D, K, e, idx, codes, dmin, codes_lp,
                       lp_relu, clip_rows, rows_per_clip);
    return nsg_check_launch("vq_combine_kernel");
}

extern "C" {

size_t nsg_vq_workspace_bytes(int64_t N, int32_t D, int32_t K)
{
    if (N < 0 || K < 0) return 0;
    return nsg_align_up((size_t)N * sizeof(float), 256) + nsg_align_up((size_t)K * sizeof(float), 256) + nsg_vq_slice_bytes(N, D, K);
}

int nsg_rowsumsq(const float *v, int64_t rows, int32_t D, float *out, void *stream)
{
    NSG_REQUIRE(v && out && rows >= 0 && D > 0, NSG_E_INVALID, "nsg_rowsumsq: bad argument");
    if (rows == 0) return NSG_OK;
    const int64_t nb = nsg_cdiv(rows * 8, 256);
    NSG_REQUIRE(nb <= 0x7fffffff, NSG_E_UNSUPPORTED, "nsg_rowsumsq: too many rows");
    hipLaunchKernelGGL(rowsumsq_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, v, rows, D, out);
    return nsg_check_launch("rowsumsq_kernel");
}

static int vq_forward_impl(bool mfma, const float *x, const float *e, int64_t N, int32_t D, int32_t K, int64_t *idx_out,
                           float *codes_out, float *dmin_out, void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && e && idx_out && N >= 0 && D > 0 && K > 0, NSG_E_INVALID, "nsg_vq_forward: bad argument");
    NSG_REQUIRE(D <= 256, NSG_E_UNSUPPORTED, "nsg_vq_forward: D=%d > 256 is not supported", D);
    if (N == 0) return NSG_OK;
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_vq_workspace_bytes(N, D, K), NSG_E_WORKSPACE,
                "nsg_vq_forward: workspace too small");
    float *x2 = reinterpret_cast<float *>(workspace);
    float *c2 = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + nsg_align_up((size_t)N * sizeof(float), 256));
    float *part_d = nullptr;
    int *part_i = nullptr;
    if (nsg_vq_slices(N, D, K) > 1) {
        char *base = reinterpret_cast<char *>(workspace) + nsg_align_up((size_t)N * sizeof(float), 256) + nsg_align_up((size_t)K * sizeof(float), 256);
        part_d = reinterpret_cast<float *>(base);
        part_i = reinterpret_cast<int *>(base + nsg_vq_slice_bytes(N, D, K) / 2);
    }
    int rc = nsg_rowsumsq(x, N, D, x2, stream);
    if (rc) return rc;
    rc = nsg_rowsumsq(e, K, D, c2, stream);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (D <= 16) return launch_vq<16>(mfma, x, e, x2, c2, N, D, K, idx_out, codes_out, dmin_out, part_d, part_i, s);
    if (D <= 32) return launch_vq<32>(mfma, x, e, x2, c2, N, D, K, idx_out, codes_out, dmin_out, part_d, part_i, s);
    if (D <= 64) return launch_vq<64>(mfma, x, e, x2, c2, N, D, K, idx_out, codes_out, dmin_out, part_d, part_i, s);
    if (D <= 128) return launch_vq<128>(mfma, x, e, x2, c2, N, D, K, idx_out, codes_out, dmin_out, part_d, part_i, s);
    return launch_vq<256>(mfma, x, e, x2, c2, N, D, K, idx_out, codes_out, dmin_out, part_d, part_i, s);
}

int nsg_vq_forward(const float *x, const float *e, int64_t N, int32_t D, int32_t K, int64_t *idx_out, float *codes_out,
                   float *dmin_out, void *workspace, size_t workspace_bytes, void *stream)
{
    return vq_forward_impl(true, x, e, N, D, K, idx_out, codes_out, dmin_out, workspace, workspace_bytes, stream);
}

int nsg_debug_vq_forward_valu(const float *x, const float *e, int64_t N, int32_t D, int32_t K, int64_t *idx_out,
                                      float *codes_out, float *dmin_out, void *workspace, size_t workspace_bytes,
                                      void *stream)
{
    return vq_forward_impl(false, x, e, N, D, K, idx_out, codes_out, dmin_out, workspace, workspace_bytes, stream);
}

int nsg_debug_dot(const float *x, const float *e, int32_t N, int32_t D, int32_t K, int32_t mode, float *out, void *stream)
{
    NSG_REQUIRE(x && e && out && N > 0 && K > 0 && D > 0 && D <= 256 && N % 32 == 0 && K % 32 == 0, NSG_E_INVALID,
                "nsg_debug_dot: bad argument");
    dim3 grid(N / 32, K / 32);
    if (mode == 1) hipLaunchKernelGGL((debug_dot_kernel<true>), grid, dim3(64), 0, (hipStream_t)stream, x, e, N, D, K, out);
    else           hipLaunchKernelGGL((debug_dot_kernel<false>), grid, dim3(64), 0, (hipStream_t)stream, x, e, N, D, K, out);
    return nsg_check_launch("debug_dot_kernel");
}

size_t nsg_index_add_workspace_bytes(int64_t N, int32_t D, int32_t K)
{
    if (N <= 0 || D <= 0 || K <= 0) return 0;
    return nsg_align_up((size_t)K * sizeof(int), 256) + nsg_wgrad_workspace_bytes(N, 1, K, D);
}

static int index_add_impl(int onehot_mode, const int64_t *idx, const float *g, int64_t N, int32_t D, int32_t K, float *out,
                          float *counts_out, void *workspace, size_t workspace_bytes, void *stream);

int nsg_index_add_rows(const int64_t *idx, const float *g, int64_t N, int32_t D, int32_t K, float *out, float *counts_out,
                       void *workspace, size_t workspace_bytes, void *stream)
{
    return index_add_impl(1, idx, g, N, D, K, out, counts_out, workspace, workspace_bytes, stream);
}

int nsg_index_add_rows_bf16x2(const int64_t *idx, const float *g, int64_t N, int32_t D, int32_t K, float *out, float *counts_out,
                              void *workspace, size_t workspace_bytes, void *stream)
{
    // rows narrower than 64 channels / not a multiple of 8 go through the fp32 kernel (same results up to rounding)
    return index_add_impl((D % 8 == 0 && D > 32) ? 2 : 1, idx, g, N, D, K, out, counts_out, workspace, workspace_bytes, stream);
}

static int index_add_impl(int onehot_mode, const int64_t *idx, const float *g, int64_t N, int32_t D, int32_t K, float *out,
                          float *counts_out, void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(idx && g && out && N > 0 && D > 0 && K > 0, NSG_E_INVALID, "nsg_index_add_rows: bad argument");
    NSG_REQUIRE(N < 0x7fffffff, NSG_E_UNSUPPORTED, "nsg_index_add_rows: too many rows");
    NSG_REQUIRE(D % 4 == 0, NSG_E_UNSUPPORTED, "nsg_index_add_rows: D=%d must be a multiple of 4", D);
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_index_add_workspace_bytes(N, D, K), NSG_E_WORKSPACE,
                "nsg_index_add_rows: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t cnt_bytes = nsg_align_up((size_t)K * sizeof(int), 256);
    if (counts_out) {
        int *cnt = reinterpret_cast<int *>(workspace);
        hipError_t err = hipMemsetAsync(cnt, 0, (size_t)K * sizeof(int), s);
        if (err != hipSuccess) return nsg_fail((int)err, "nsg_index_add_rows: memset failed");
        const int nb = (int)(nsg_cdiv(N, 1024) > 256 ? 256 : nsg_cdiv(N, 1024));
        hipLaunchKernelGGL(count_codes_kernel, dim3(nb), dim3(256), K <= COUNT_LDS_BINS ? (size_t)K * sizeof(int) : 0, s, idx, N, K, cnt);
        hipLaunchKernelGGL(counts_to_float_kernel, dim3((K + 255) / 256), dim3(256), 0, s, cnt, K, counts_out);
        int rc = nsg_check_launch("count_codes_kernel");
        if (rc) return rc;
    }
    WgradParams p = {};
    p.P = nullptr;
    p.Q = g;
    p.idx = idx;
    p.B = 1; p.PH = 1; p.PW = (int)N; p.A = K;
    p.QH = 1; p.QW = (int)N; p.C = D;
    p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
    p.Mp = (int)N;
    p.onehot = onehot_mode;
    return nsg_launch_wgrad(p, out, reinterpret_cast<char *>(workspace) + cnt_bytes, workspace_bytes - cnt_bytes, s);
}

int nsg_gather_rows(const float *e, const int64_t *idx, int64_t N, int32_t D, int32_t K, float *out, void *stream)
{
    NSG_REQUIRE(e && idx && out && N >= 0 && D > 0 && K > 0, NSG_E_INVALID, "nsg_gather_rows: bad argument");
    if (N == 0) return NSG_OK;
    const int vec = (D % 4 == 0) && nsg_aligned16(e) && nsg_aligned16(out);
    const int64_t work = vec ? N * (D / 4) : N * (int64_t)D;
    const int nb = (int)(nsg_cdiv(work, 256) > 4096 ? 4096 : nsg_cdiv(work, 256));
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, e, idx, N, D, K, out, vec);
    return nsg_check_launch("gather_rows_kernel");
}

int nsg_vq_ema_update(float *e, float *ema_n, float *ema_s, const float *n, const float *s, int32_t K, int32_t D,
                      float decay, float eps, float *scratch, void *stream)
{
    NSG_REQUIRE(e && ema_n && ema_s && n && s && K > 0 && D > 0, NSG_E_INVALID, "nsg_vq_ema_update: bad argument");
    NSG_REQUIRE(scratch, NSG_E_INVALID, "nsg_vq_ema_update: scratch (1 float) required");
    hipLaunchKernelGGL(ema_counts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ema_n, n, K, decay, scratch);
    const int64_t tot = (int64_t)K * D;
    const int nb = (int)(nsg_cdiv(tot, 256) > 2048 ? 2048 : nsg_cdiv(tot, 256));
    hipLaunchKernelGGL(ema_codes_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, e, ema_n, ema_s, s, K, D, decay, eps, scratch);
    return nsg_check_launch("ema_update");
}

}  // extern "C"
