// BatchNorm2d over NHWC activations viewed as [M][C] (M = B*H*W).  Reference call sites:
// src/models.py:151,154,166,180 (nn.BatchNorm2d, training and eval).  HBM-bound kernels:
// 16-byte accesses, each row of C channels is contiguous so a wave reads whole 128-B lines.
// Activations may be stored fp32 or bf16 (templates below); all arithmetic, statistics and
// per-channel parameters are fp32.
//
// Statistics are formed per slab of rows as (count, mean, M2 about the slab mean) in one pass about a
// pivot row, and pooled in double in slab order: no E[x^2]-E[x]^2 cancellation and bitwise
// reproducible (no atomics).
// The backward of a BatchNorm that was followed by a fused ReLU re-derives the ReLU mask from x itself
// ((x-mean)*(invstd*gamma)+beta > 0, the forward's own expression) instead of reading the stored output.
#include "nsg_common.h"

namespace {

constexpr int MAX_SLABS = 1024;

struct SlabGeom {
    int nslab;
    int rows;
};
inline SlabGeom slab_geom(int64_t M)
{
    SlabGeom g;
    int64_t n = nsg_cdiv(M, 64);   // >= 64 rows per slab (256 left 320 blocks for the 82 K rows of BASELINE configs[3]: 1.25 per CU)
    if (n > MAX_SLABS) n = MAX_SLABS;
    if (n < 1) n = 1;
    g.rows = (int)nsg_cdiv(M, n);
    g.nslab = (int)nsg_cdiv(M, g.rows);
    return g;
}

// threads: cg = tid % CW owns channels W*cg .. W*cg+W-1, rg = tid / CW strides over the slab's rows.
// partial[slab] = { mean[C], M2[C] } (count is implied by the slab geometry).
// ONE pass: sums of d = v - pivot and d^2 with pivot = the slab's first row (a sample of the column, so
// |mean - pivot| is of the order of the column's spread and M2 = S2 - S1^2/n loses no more than a few bits).
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const T *__restrict__ x, int64_t M, int C, int slab_rows,
                                                               float *__restrict__ partial)
{
    constexpr int W = Elem<T>::N;
    __shared__ float red[2 * 256 * W];
    const int CW = C / W;
    const int rgroups = 256 / CW;
    const int tid = threadIdx.x;
    const int cg = tid % CW, rg = tid / CW;
    const bool active = rg < rgroups;
    const int64_t r0 = (int64_t)blockIdx.x * slab_rows;
    const int64_t r1 = min(M, r0 + slab_rows);
    const int n = (int)(r1 - r0);

    float pv[W];
    if (active) {
        float s1[W], s2[W];
        ldw<T, W>(x + r0 * C + cg * W, pv);
#pragma unroll
        for (int e = 0; e < W; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll 4
        for (int64_t r = r0 + rg; r < r1; r += rgroups) {
            float v[W];
            ldw<T, W>(x + r * C + cg * W, v);
#pragma unroll
            for (int e = 0; e < W; ++e) { const float d = v[e] - pv[e]; s1[e] += d; s2[e] += d * d; }
        }
#pragma unroll
        for (int e = 0; e < W; ++e) { red[(rg * CW + cg) * W + e] = s1[e]; red[256 * W + (rg * CW + cg) * W + e] = s2[e]; }
    }
    __syncthreads();
    if (tid < CW) {
        float *dst = partial + (size_t)blockIdx.x * 2 * C;
        const float inv_n = 1.f / (float)n;
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float t1 = 0.f, t2 = 0.f;
            for (int g = 0; g < rgroups; ++g) { t1 += red[(g * CW + tid) * W + e]; t2 += red[256 * W + (g * CW + tid) * W + e]; }
            dst[tid * W + e] = pv[e] + t1 * inv_n;
            dst[C + tid * W + e] = fmaxf(t2 - t1 * t1 * inv_n, 0.f);
        }
    }
}

// Fixed-shape sum of one double per thread over the 256 threads of a block (deterministic); result in every thread.
// Xor-butterflies inside each wave (no barrier), then the 4 wave totals through LDS in wave order.
__device__ __forceinline__ double block_sum256(double v, double *red, int tid)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const double r = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    return r;
}

// One block per 4 channels; thread j takes slabs j, j+256, ... (16-byte loads of the 4 channels' partials, all issued
// before any arithmetic: one memory latency), then two fixed-shape tree sums per channel in double:
//   n = sum n_s,  mean = sum n_s m_s / n,  M2 = sum ( q_s + n_s (m_s - mean)^2 )       (the pooled-variance identity)
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float *__restrict__ partial, int nslab, int slab_rows, int64_t M,
                                                             int C, float eps, float momentum, float *mean, float *invstd,
                                                             float *running_mean, float *running_var)
{
    constexpr int PER = MAX_SLABS / 256;
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * 4;
    v4f ms[PER], qs[PER];
    float ns[PER];
    float rm0[4], rv0[4];        // the running statistics travel with the partials (read by every thread: no lane condition on a load)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int cc = c0 + e < C ? c0 + e : C - 1;
        rm0[e] = running_mean ? running_mean[cc] : 0.f;
        rv0[e] = running_var ? running_var[cc] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int s = tid + 256 * i;
        const int sc = s < nslab ? s : nslab - 1;                     // clamped, unconditional loads
        ms[i] = *reinterpret_cast<const v4f *>(partial + (size_t)sc * 2 * C + c0);
        qs[i] = *reinterpret_cast<const v4f *>(partial + (size_t)sc * 2 * C + C + c0);
        const int64_t r0 = (int64_t)sc * slab_rows;
        ns[i] = s < nslab ? (float)(min(M, r0 + slab_rows) - r0) : 0.f;
    }
    double mu[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) t += (double)ns[i] * (double)ms[i][e];
        mu[e] = block_sum256(t, red, tid) / (double)M;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const double dl = (double)ms[i][e] - mu[e];
            t += ns[i] > 0.f ? (double)qs[i][e] + (double)ns[i] * dl * dl : 0.0;
        }
        const double m2 = block_sum256(t, red, tid);
        if (tid == 0 && c0 + e < C) {
            const int c = c0 + e;
            const double var_b = m2 / (double)M;
            mean[c] = (float)mu[e];
            invstd[c] = (float)(1.0 / sqrt(var_b + (double)eps));
            if (running_mean) running_mean[c] = (1.f - momentum) * rm0[e] + momentum * (float)mu[e];
            if (running_var) {
                const double var_u = M > 1 ? m2 / (double)(M - 1) : var_b;
                running_var[c] = (1.f - momentum) * rv0[e] + momentum * (float)var_u;
            }
        }
    }
}

// tiles: [ntiles][3][C] = (count n_t, sum S_t, M2_t about the tile mean), written by a producer's epilogue / statistics pass.
// With N = sum n_t, S = sum S_t:  mean = S/N,  M2 = sum_t ( M2_t + S_t^2/n_t ) - S^2/N  (double).
// Two stages, both with a thread per channel (coalesced rows of the tile records): TILE_CHUNKS blocks each fold a
// contiguous run of tiles in tile order into (N, S, Q) per channel, then one block folds the chunks in chunk order.
constexpr int TILE_CHUNKS = 64;
__global__ __launch_bounds__(256) void bn_stats_tiles_fold_kernel(const float *__restrict__ tiles, int ntiles, int C, double *__restrict__ chunks)
{
    __shared__ double red[3][256];
    const int tid = threadIdx.x;
    const int cx = tid & 63, ty = tid >> 6;
    const int per = (ntiles + TILE_CHUNKS - 1) / TILE_CHUNKS;
    const int t0 = blockIdx.x * per, t1 = min(ntiles, t0 + per);
    for (int cb = 0; cb < C; cb += 64) {
        const int c = cb + cx;
        double N = 0.0, S = 0.0, Q = 0.0;
        if (c < C)
            for (int t = t0 + ty; t < t1; t += 4) {
                const float *rec = tiles + (size_t)t * 3 * C + c;
                const double n = rec[0];
                if (n == 0.0) continue;
                const double st = rec[C], qt = rec[2 * C];
                N += n;
                S += st;
                Q += qt + st * st / n;
            }
        red[0][tid] = N; red[1][tid] = S; red[2][tid] = Q;
        __syncthreads();
        if (ty == 0 && c < C) {
            double *dst = chunks + (size_t)blockIdx.x * 3 * C;
            dst[c] = ((red[0][cx] + red[0][64 + cx]) + red[0][128 + cx]) + red[0][192 + cx];
            dst[C + c] = ((red[1][cx] + red[1][64 + cx]) + red[1][128 + cx]) + red[1][192 + cx];
            dst[2 * C + c] = ((red[2][cx] + red[2][64 + cx]) + red[2][128 + cx]) + red[2][192 + cx];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void bn_stats_tiles_final_kernel(const double *__restrict__ chunks, int64_t M, int C, float eps, float momentum,
                                                                   float *mean, float *invstd, float *running_mean, float *running_var)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rm0 = running_mean ? running_mean[c] : 0.f, rv0 = running_var ? running_var[c] : 0.f;     // (in flight with the chunks)
    double N = 0.0, S = 0.0, Q = 0.0;
    for (int k0 = 0; k0 < TILE_CHUNKS; k0 += 16) {        // 48 loads in flight, added in chunk order
        double n[16], sv[16], q[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            n[k] = chunks[(size_t)(k0 + k) * 3 * C + c];
            sv[k] = chunks[(size_t)(k0 + k) * 3 * C + C + c];
            q[k] = chunks[(size_t)(k0 + k) * 3 * C + 2 * C + c];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) { N += n[k]; S += sv[k]; Q += q[k]; }
    }
    const double mu = S / N;
    double m2 = Q - S * S / N;
    if (m2 < 0.0) m2 = 0.0;
    const double var_b = m2 / (double)M;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var_b + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * rm0 + momentum * (float)mu;
    if (running_var) {
        const double var_u = M > 1 ? m2 / (double)(M - 1) : var_b;
        running_var[c] = (1.f - momentum) * rv0 + momentum * (float)var_u;
    }
}

__global__ void bn_eval_stats_kernel(const float *rm, const float *rv, int C, float eps, float *mean, float *invstd)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        mean[c] = rm[c];
        invstd[c] = 1.0f / sqrtf(rv[c] + eps);
    }
}

// y = (x-mean)*invstd*gamma + beta  [relu]  [+ (relu) residual];   x/residual of type TX, y of type TY.
// The grid stride is a multiple of the row length (launcher), so a thread keeps its W channels: their parameters are
// loaded once, not per element (the kernel was instruction-bound on those loads: 3.7 TB/s on tensors beyond the MALL).
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void bn_apply_kernel(const TX *__restrict__ x, const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, const TX *__restrict__ residual,
                                                       TY *__restrict__ y, int64_t nw, int CW, int relu, int relu_res)
{
    // relu bit 0: ReLU right after the affine map; bit 1: ReLU of the final value (after the residual add)
    const int relu_bn = relu & 1, relu_out = relu & 2;
    constexpr int W = Width<TX, TY>::W;
    const int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int c = (int)(i0 % CW) * W;
    float mu[W], sc[W], be[W];
#pragma unroll
    for (int e = 0; e < W; ++e) { mu[e] = mean[c + e]; sc[e] = invstd[c + e] * gamma[c + e]; be[e] = beta[c + e]; }
    for (int64_t i = i0; i < nw; i += (int64_t)gridDim.x * blockDim.x) {
        float xv[W], o[W];
        ldw<TX, W>(x + i * W, xv);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float t = (xv[e] - mu[e]) * sc[e] + be[e];
            if (relu_bn) t = fmaxf(t, 0.f);
            o[e] = t;
        }
        if (residual) {
            float rv[W];
            ldw<TX, W>(residual + i * W, rv);
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] += relu_res ? fmaxf(rv[e], 0.f) : rv[e];
        }
        if (relu_out)
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] = fmaxf(o[e], 0.f);
        stw<TY, W>(y + i * W, o);
    }
}

// per-slab partial sums of dyh and dyh*xhat (dyh = dy masked by the fused ReLU)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T *__restrict__ x, const T *__restrict__ yrelu,
                                                             const T *__restrict__ dy, const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                             const float *__restrict__ relu_beta, int64_t M, int C,
                                                             int slab_rows, float *__restrict__ partial)
{
    constexpr int W = Elem<T>::N;
    __shared__ float red[2 * 256 * W];
    const int CW = C / W;
    const int rgroups = 256 / CW;
    const int tid = threadIdx.x;
    const int cg = tid % CW, rg = tid / CW;
    const bool active = rg < rgroups;
    const int64_t r0 = (int64_t)blockIdx.x * slab_rows;
    const int64_t r1 = min(M, r0 + slab_rows);
    if (active) {
        float s1[W], s2[W], mu[W], is[W], sc[W], be[W];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            s1[e] = 0.f; s2[e] = 0.f; mu[e] = mean[cg * W + e]; is[e] = invstd[cg * W + e];
            sc[e] = relu_beta ? is[e] * gamma[cg * W + e] : 0.f;
            be[e] = relu_beta ? relu_beta[cg * W + e] : 0.f;
        }
        for (int64_t r = r0 + rg; r < r1; r += rgroups) {
            const size_t o = (size_t)r * C + cg * W;
            float g[W], xv[W];
            ldw<T, W>(dy + o, g);
            ldw<T, W>(x + o, xv);
            if (relu_beta) {          // the forward's own expression (bn_apply_kernel): same rounding, same decision
#pragma unroll
                for (int e = 0; e < W; ++e) g[e] = ((xv[e] - mu[e]) * sc[e] + be[e]) > 0.f ? g[e] : 0.f;
            } else if (yrelu) {
                float yv[W];
                ldw<T, W>(yrelu + o, yv);
#pragma unroll
                for (int e = 0; e < W; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < W; ++e) { s1[e] += g[e]; s2[e] += g[e] * ((xv[e] - mu[e]) * is[e]); }
        }
#pragma unroll
        for (int e = 0; e < W; ++e) { red[(rg * CW + cg) * W + e] = s1[e]; red[256 * W + (rg * CW + cg) * W + e] = s2[e]; }
    }
    __syncthreads();
    if (tid < CW) {
        float *dst = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float t1 = 0.f, t2 = 0.f;
            for (int g = 0; g < rgroups; ++g) { t1 += red[(g * CW + tid) * W + e]; t2 += red[256 * W + (g * CW + tid) * W + e]; }
            dst[tid * W + e] = t1;
            dst[C + tid * W + e] = t2;
        }
    }
}

// dbeta[c] = sum_s partial[s][c], dgamma[c] = sum_s partial[s][C + c]: one block per 4 channels, tree sums in double
// colsum_partial != null: out3[c] = sum_s colsum_partial[s][c] in the same launch (slab_sum_final_kernel's job: a producer that
// leaves both kinds of partials, the fused 1x1 backward, pays one finaliser launch instead of two)
__device__ __forceinline__ void bn_bwd_final_body(int bid, const float *__restrict__ partial, int nslab, int C, float *dgamma, float *dbeta,
                                                  const float *__restrict__ colsum_partial, float *out3)
{
    constexpr int PER = MAX_SLABS / 256;
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int c0 = bid * 4;
    v4f a[PER], b[PER], c3[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int s = tid + 256 * i;
        const int sc = s < nslab ? s : nslab - 1;
        a[i] = *reinterpret_cast<const v4f *>(partial + (size_t)sc * 2 * C + c0);
        b[i] = *reinterpret_cast<const v4f *>(partial + (size_t)sc * 2 * C + C + c0);
        c3[i] = colsum_partial ? *reinterpret_cast<const v4f *>(colsum_partial + (size_t)sc * C + c0) : v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const bool ok = tid + 256 * i < nslab;
            t1 += ok ? (double)a[i][e] : 0.0;
            t2 += ok ? (double)b[i][e] : 0.0;
            t3 += ok ? (double)c3[i][e] : 0.0;
        }
        const double s1 = block_sum256(t1, red, tid);
        const double s2 = block_sum256(t2, red, tid);
        const double s3 = colsum_partial ? block_sum256(t3, red, tid) : 0.0;
        if (tid == 0 && c0 + e < C) {
            dbeta[c0 + e] = (float)s1;
            dgamma[c0 + e] = (float)s2;
            if (colsum_partial) out3[c0 + e] = (float)s3;
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float *__restrict__ partial, int nslab, int C, float *dgamma, float *dbeta,
                                                           const float *__restrict__ colsum_partial, float *out3)
{
    bn_bwd_final_body(blockIdx.x, partial, nslab, C, dgamma, dbeta, colsum_partial, out3);
}

// bn_bwd_final_kernel in blocks 0 .. C / 4 - 1 and, in the blocks behind them, wdst[e] = sum over slabs (slab order) of
// wpartial[slab][e], e < wn -- wgrad_reduce_kernel for one tap, its sharing of the slabs among 8 lanes and its order of adds: the
// two finalisers of the fused 1x1 backward (BatchNorm sums in front, the weight gradient's block partials) in ONE launch (a
// dependent launch costs 4.7 us here whatever it does).
__global__ __launch_bounds__(256) void bn_bwd_final_wreduce_kernel(const float *__restrict__ partial, int nslab, int C, float *dgamma, float *dbeta,
                                                                   const float *__restrict__ colsum_partial, float *out3,
                                                                   const float *__restrict__ wpartial, float *__restrict__ wdst, int wn, int wsplit)
{
    const int nbn = C / 4;
    if ((int)blockIdx.x < nbn) {
        bn_bwd_final_body(blockIdx.x, partial, nslab, C, dgamma, dbeta, colsum_partial, out3);
        return;
    }
    __shared__ float wred[256];
    const int per_block = 256 / wsplit;
    const int tid = threadIdx.x;
    const int sub = tid / per_block, loc = tid - sub * per_block;
    const int e = ((int)blockIdx.x - nbn) * per_block + loc;
    float sacc = 0.f;
    if (e < wn) {
        const int chunk = (nslab + wsplit - 1) / wsplit;
        const int s0 = sub * chunk, s1 = min(nslab, s0 + chunk);
        if (s1 > s0) sacc = nsg_strided_sum<float>(wpartial + (size_t)s0 * wn + e, (size_t)wn, s1 - s0);
    }
    if (wsplit > 1) {
        wred[tid] = sacc;
        __syncthreads();
        if (sub == 0) {
            sacc = 0.f;
            for (int k = 0; k < wsplit; ++k) sacc += wred[k * per_block + loc];
        }
    }
    if (sub == 0 && e < wn) wdst[e] = sacc;
}

// dx = gamma*invstd*(dyh - mean(dyh) - xhat*mean(dyh*xhat)), slab-structured; optionally also the per-slab
// column sums of dx (= bias gradient of the convolution in front of this BatchNorm) into `partial`
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T *__restrict__ x, const T *__restrict__ yrelu,
                                                           const T *__restrict__ dy, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                           const float *__restrict__ relu_beta, T *__restrict__ dx, int64_t M, int C,
                                                           int slab_rows, float inv_m, float *__restrict__ partial)
{
    constexpr int W = Elem<T>::N;
    __shared__ float red[256 * W];
    const int CW = C / W;
    const int rgroups = 256 / CW;
    const int tid = threadIdx.x;
    const int cg = tid % CW, rg = tid / CW;
    const bool active = rg < rgroups;
    const int64_t r0 = (int64_t)blockIdx.x * slab_rows;
    const int64_t r1 = min(M, r0 + slab_rows);
    float s[W];
#pragma unroll
    for (int e = 0; e < W; ++e) s[e] = 0.f;
    if (active) {
        float mu[W], is[W], sc[W], dg[W], db[W], fs[W], be[W];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const int c = cg * W + e;
            mu[e] = mean[c]; is[e] = invstd[c]; sc[e] = gamma[c] * is[e]; dg[e] = dgamma[c] * inv_m; db[e] = dbeta[c] * inv_m;
            fs[e] = is[e] * gamma[c];                  // the forward's scale, in the forward's operand order
            be[e] = relu_beta ? relu_beta[c] : 0.f;
        }
        for (int64_t r = r0 + rg; r < r1; r += rgroups) {
            const size_t o = (size_t)r * C + cg * W;
            float g[W], xv[W], d[W];
            ldw<T, W>(dy + o, g);
            ldw<T, W>(x + o, xv);
            if (relu_beta) {
#pragma unroll
                for (int e = 0; e < W; ++e) g[e] = ((xv[e] - mu[e]) * fs[e] + be[e]) > 0.f ? g[e] : 0.f;
            } else if (yrelu) {
                float yv[W];
                ldw<T, W>(yrelu + o, yv);
#pragma unroll
                for (int e = 0; e < W; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < W; ++e) {
                d[e] = sc[e] * (g[e] - db[e] - ((xv[e] - mu[e]) * is[e]) * dg[e]);
                s[e] += d[e];
            }
            stw<T, W>(dx + o, d);
        }
    }
    if (partial == nullptr) return;
    if (active)
#pragma unroll
        for (int e = 0; e < W; ++e) red[(rg * CW + cg) * W + e] = s[e];
    __syncthreads();
    if (tid < CW) {
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float t = 0.f;
            for (int g = 0; g < rgroups; ++g) t += red[(g * CW + tid) * W + e];
            partial[(size_t)blockIdx.x * C + tid * W + e] = t;
        }
    }
}

// out[c] = sum_s partial[s][c]: one block per 4 channels
__global__ __launch_bounds__(256) void slab_sum_final_kernel(const float *__restrict__ partial, int nslab, int C, float *out)
{
    constexpr int PER = MAX_SLABS / 256;
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * 4;
    v4f a[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int s = tid + 256 * i;
        a[i] = *reinterpret_cast<const v4f *>(partial + (size_t)(s < nslab ? s : nslab - 1) * C + c0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) t += (tid + 256 * i < nslab) ? (double)a[i][e] : 0.0;
        const double sm = block_sum256(t, red, tid);
        if (tid == 0 && c0 + e < C) out[c0 + e] = (float)sm;
    }
}

inline int ew_blocks(int64_t n) { return (int)(nsg_cdiv(n, 256) > 4096 ? 4096 : (nsg_cdiv(n, 256) < 1 ? 1 : nsg_cdiv(n, 256))); }

inline int check_mc(const char *fn, int64_t M, int C, int dtype)
{
    if (M <= 0 || C <= 0) return nsg_fail(NSG_E_INVALID, "%s: bad size", fn);
    if (dtype != NSG_F32 && dtype != NSG_BF16) return nsg_fail(NSG_E_INVALID, "%s: unknown dtype %d", fn, dtype);
    const int w = dtype == NSG_BF16 ? 8 : 4;
    if (C % w != 0 || C > 1024) return nsg_fail(NSG_E_UNSUPPORTED, "%s: C=%d must be a multiple of %d and <= 1024", fn, C, w);
    if (M * (int64_t)C >= (1ll << 31)) return nsg_fail(NSG_E_UNSUPPORTED, "%s: tensor too large", fn);
    return NSG_OK;
}

}  // namespace

// The same in ONE launch for up to MAX_SLABS records (the flat 1x1 GEMM writes 256-512): one block per 4 channels, thread j takes
// records j, j + 256, ... (all loads first), three fixed-shape tree sums in double: N, S, T = sum (Q_t + S_t^2 / n_t).
__global__ __launch_bounds__(256) void bn_stats_tiles_onepass_kernel(const float *__restrict__ tiles, int ntiles, int64_t M, int C, float eps,
                                                                     float momentum, float *mean, float *invstd, float *running_mean,
                                                                     float *running_var)
{
    constexpr int PER = MAX_SLABS / 256;
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * 4;
    v4f nv[PER], sv[PER], qv[PER];
    float rm0[4], rv0[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        rm0[e] = running_mean ? running_mean[c0 + e] : 0.f;
        rv0[e] = running_var ? running_var[c0 + e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int t = tid + 256 * i;
        const int tc = t < ntiles ? t : ntiles - 1;                  // clamped, unconditional loads; masked below
        const float *rec = tiles + (size_t)tc * 3 * C + c0;
        nv[i] = *reinterpret_cast<const v4f *>(rec);
        sv[i] = *reinterpret_cast<const v4f *>(rec + C);
        qv[i] = *reinterpret_cast<const v4f *>(rec + 2 * C);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double n = 0.0, sm = 0.0, tq = 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const bool live = tid + 256 * i < ntiles;
            const double ni = live ? (double)nv[i][e] : 0.0, si = live ? (double)sv[i][e] : 0.0, qi = live ? (double)qv[i][e] : 0.0;
            n += ni;
            sm += si;
            tq += ni > 0.0 ? qi + si * si / ni : 0.0;
        }
        const double N = block_sum256(n, red, tid), S = block_sum256(sm, red, tid), T = block_sum256(tq, red, tid);
        if (tid == 0) {
            const int c = c0 + e;
            const double mu = S / N;
            double m2 = T - S * S / N;
            if (m2 < 0.0) m2 = 0.0;
            const double var_b = m2 / (double)M;
            mean[c] = (float)mu;
            invstd[c] = (float)(1.0 / sqrt(var_b + (double)eps));
            if (running_mean) running_mean[c] = (1.f - momentum) * rm0[e] + momentum * (float)mu;
            if (running_var) {
                const double var_u = M > 1 ? m2 / (double)(M - 1) : var_b;
                running_var[c] = (1.f - momentum) * rv0[e] + momentum * (float)var_u;
            }
        }
    }
}

// used by nsg_conv_forward_bnstats (conv_api.hip): merge the per-tile statistics the conv epilogue wrote
int nsg_bn_stats_from_tiles(const float *tiles, int ntiles, int64_t M, int C, float eps, float momentum, float *mean,
                            float *invstd, float *running_mean, float *running_var, hipStream_t s)
{
    if (ntiles <= MAX_SLABS && C % 4 == 0) {
        hipLaunchKernelGGL(bn_stats_tiles_onepass_kernel, dim3(C / 4), dim3(256), 0, s, tiles, ntiles, M, C, eps, momentum, mean, invstd,
                           running_mean, running_var);
        return nsg_check_launch("bn_stats_tiles_onepass_kernel");
    }
    // the chunk records live behind the tile records (callers size the tile buffer with nsg_bn_tiles_bytes)
    double *chunks = reinterpret_cast<double *>(const_cast<float *>(tiles) + nsg_align_up((size_t)ntiles * 3 * C, 64));
    hipLaunchKernelGGL(bn_stats_tiles_fold_kernel, dim3(TILE_CHUNKS), dim3(256), 0, s, tiles, ntiles, C, chunks);
    hipLaunchKernelGGL(bn_stats_tiles_final_kernel, dim3((C + 255) / 256), dim3(256), 0, s, chunks, M, C, eps, momentum, mean, invstd,
                       running_mean, running_var);
    return nsg_check_launch("bn_stats_from_tiles");
}

// bytes of a tile-statistics buffer for ntiles records of C channels: the records + the finalizer's chunk records
size_t nsg_bn_tiles_bytes(int64_t ntiles, int C)
{
    return nsg_align_up((size_t)ntiles * 3 * C, 64) * sizeof(float) + (size_t)TILE_CHUNKS * 3 * C * sizeof(double) + 256;
}

// the slab structure of nsg_bn_backward_sums over M rows: a producer of dy that forms the same sums itself (elementwise.hip:
// nsg_vq_losses_indexed_bn) walks the rows the same way, so its results are those of the separate pass, bit for bit
void nsg_bn_slab_geom(int64_t M, int *nslab, int *rows)
{
    const SlabGeom g = slab_geom(M);
    *nslab = g.nslab;
    *rows = g.rows;
}

int nsg_launch_bn_bwd_final(const float *partial, int nslab, int C, float *dgamma, float *dbeta, hipStream_t s)
{
    if (nslab < 1 || nslab > MAX_SLABS || C % 4) return nsg_fail(NSG_E_INVALID, "bn_bwd_final: %d slabs / %d channels not supported", nslab, C);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C / 4), dim3(256), 0, s, partial, nslab, C, dgamma, dbeta, (const float *)nullptr, (float *)nullptr);
    return nsg_check_launch("bn_bwd_final_kernel");
}

// ... and the column sums of a second partial array [nslab][C] in the same launch
int nsg_launch_bn_bwd_final_colsum(const float *partial, const float *colsum_partial, int nslab, int C, float *dgamma, float *dbeta,
                                   float *colsum, hipStream_t s)
{
    if (nslab < 1 || nslab > MAX_SLABS || C % 4) return nsg_fail(NSG_E_INVALID, "bn_bwd_final: %d slabs / %d channels not supported", nslab, C);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C / 4), dim3(256), 0, s, partial, nslab, C, dgamma, dbeta, colsum_partial, colsum);
    return nsg_check_launch("bn_bwd_final_kernel");
}

// ... and the slab sum of the weight gradient's block partials wpartial [nslab][wn] -> wdst [wn] (one tap: nsg_launch_wgrad_reduce's result, bit for bit)
int nsg_launch_bn_bwd_final_wreduce(const float *partial, const float *colsum_partial, int nslab, int C, float *dgamma, float *dbeta, float *colsum,
                                    const float *wpartial, float *wdst, int wn, hipStream_t s)
{
    if (nslab < 1 || nslab > MAX_SLABS || C % 4) return nsg_fail(NSG_E_INVALID, "bn_bwd_final: %d slabs / %d channels not supported", nslab, C);
    const int split = (nslab >= 64 && wn < 512 * 256) ? 8 : 1;      // (nsg_launch_wgrad_reduce's choice)
    const int wblocks = (wn + 256 / split - 1) / (256 / split);
    hipLaunchKernelGGL(bn_bwd_final_wreduce_kernel, dim3(C / 4 + wblocks), dim3(256), 0, s, partial, nslab, C, dgamma, dbeta, colsum_partial, colsum,
                       wpartial, wdst, wn, split);
    return nsg_check_launch("bn_bwd_final_wreduce_kernel");
}

int nsg_launch_slab_sum_final(const float *partial, int nslab, int C, float *out, hipStream_t s)
{
    if (nslab < 1 || nslab > MAX_SLABS || C % 4) return nsg_fail(NSG_E_INVALID, "slab_sum_final: %d slabs / %d channels not supported", nslab, C);
    hipLaunchKernelGGL(slab_sum_final_kernel, dim3(C / 4), dim3(256), 0, s, partial, nslab, C, out);
    return nsg_check_launch("slab_sum_final_kernel");
}

extern "C" {

size_t nsg_bn_workspace_bytes(int64_t M, int32_t C)
{
    if (M <= 0 || C <= 0) return 0;
    return (size_t)slab_geom(M).nslab * 2 * C * sizeof(float);
}

int nsg_bn_stats(const void *x, int64_t M, int32_t C, int32_t dtype, float eps, float momentum, float *mean, float *invstd,
                 float *running_mean, float *running_var, void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && mean && invstd, NSG_E_INVALID, "nsg_bn_stats: null pointer");
    int rc = check_mc("nsg_bn_stats", M, C, dtype);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(x), NSG_E_INVALID, "nsg_bn_stats: x must be 16-byte aligned");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_bn_workspace_bytes(M, C), NSG_E_WORKSPACE, "nsg_bn_stats: workspace too small");
    const SlabGeom g = slab_geom(M);
    hipStream_t s = (hipStream_t)stream;
    float *partial = reinterpret_cast<float *>(workspace);
    if (dtype == NSG_BF16)
        hipLaunchKernelGGL((bn_stats_partial_kernel<bf16_t>), dim3(g.nslab), dim3(256), 0, s, reinterpret_cast<const bf16_t *>(x), M, C, g.rows, partial);
    else
        hipLaunchKernelGGL((bn_stats_partial_kernel<float>), dim3(g.nslab), dim3(256), 0, s, reinterpret_cast<const float *>(x), M, C, g.rows, partial);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C / 4), dim3(256), 0, s, partial, g.nslab, g.rows, M, C, eps,
                       momentum, mean, invstd, running_mean, running_var);
    return nsg_check_launch("bn_stats");
}

int nsg_bn_eval_stats(const float *running_mean, const float *running_var, int32_t C, float eps, float *mean,
                      float *invstd, void *stream)
{
    NSG_REQUIRE(running_mean && running_var && mean && invstd && C > 0, NSG_E_INVALID, "nsg_bn_eval_stats: bad argument");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, running_mean,
                       running_var, C, eps, mean, invstd);
    return nsg_check_launch("bn_eval_stats");
}

int nsg_bn_apply(const void *x, const float *mean, const float *invstd, const float *gamma, const float *beta,
                 const void *residual, void *y, int64_t M, int32_t C, int32_t relu, int32_t relu_residual, int32_t dtype,
                 int32_t y_dtype, void *stream)
{
    NSG_REQUIRE(x && mean && invstd && gamma && beta && y, NSG_E_INVALID, "nsg_bn_apply: null pointer");
    int rc = check_mc("nsg_bn_apply", M, C, dtype);
    if (rc) return rc;
    NSG_REQUIRE(y_dtype == NSG_F32 || y_dtype == NSG_BF16, NSG_E_INVALID, "nsg_bn_apply: unknown output dtype");
    NSG_REQUIRE(nsg_aligned16(x) && nsg_aligned16(y) && (!residual || nsg_aligned16(residual)), NSG_E_INVALID,
                "nsg_bn_apply: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int W = (dtype == NSG_BF16 || y_dtype == NSG_BF16) ? 8 : 4;
    NSG_REQUIRE(C % W == 0, NSG_E_UNSUPPORTED, "nsg_bn_apply: C=%d must be a multiple of %d", C, W);
    const int64_t nw = M * C / W;
    // grid stride (blocks * 256) must be a multiple of the row length CW = C / W so that a thread keeps its channels
    int blocks = ew_blocks(nw);
    {
        const int CW = C / W;
        int a = CW, b = 256;
        while (b) { const int t = a % b; a = b; b = t; }      // a = gcd(CW, 256)
        const int mult = CW / a;
        blocks = (blocks + mult - 1) / mult * mult;
    }
    const dim3 grid(blocks), blk(256);
#define NSG_BN_APPLY(TX, TY)                                                                                             \
    hipLaunchKernelGGL((bn_apply_kernel<TX, TY>), grid, blk, 0, s, reinterpret_cast<const TX *>(x), mean, invstd, gamma, \
                       beta, reinterpret_cast<const TX *>(residual), reinterpret_cast<TY *>(y), nw, C / W, relu, relu_residual)
    if (dtype == NSG_F32 && y_dtype == NSG_F32) NSG_BN_APPLY(float, float);
    else if (dtype == NSG_BF16 && y_dtype == NSG_BF16) NSG_BN_APPLY(bf16_t, bf16_t);
    else if (dtype == NSG_BF16 && y_dtype == NSG_F32) NSG_BN_APPLY(bf16_t, float);
    else NSG_BN_APPLY(float, bf16_t);
#undef NSG_BN_APPLY
    return nsg_check_launch("bn_apply_kernel");
}

int nsg_bn_backward(const void *x, const void *y_relu, const void *dy, const float *mean, const float *invstd,
                    const float *gamma, const float *relu_beta, void *dx, float *dgamma, float *dbeta, float *dx_colsum,
                    int64_t M, int32_t C, int32_t dtype, void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && dy && mean && invstd && gamma && dx && dgamma && dbeta, NSG_E_INVALID, "nsg_bn_backward: null pointer");
    int rc = check_mc("nsg_bn_backward", M, C, dtype);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(x) && nsg_aligned16(dy) && nsg_aligned16(dx) && (!y_relu || nsg_aligned16(y_relu)), NSG_E_INVALID,
                "nsg_bn_backward: pointers must be 16-byte aligned");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_bn_workspace_bytes(M, C), NSG_E_WORKSPACE, "nsg_bn_backward: workspace too small");
    const SlabGeom g = slab_geom(M);
    hipStream_t s = (hipStream_t)stream;
    float *partial = reinterpret_cast<float *>(workspace);
    const float inv_m = 1.0f / (float)M;
    if (dtype == NSG_BF16) {
        typedef bf16_t T;
        hipLaunchKernelGGL((bn_bwd_partial_kernel<T>), dim3(g.nslab), dim3(256), 0, s, (const T *)x, (const T *)y_relu, (const T *)dy, mean, invstd, gamma, relu_beta, M, C, g.rows, partial);
        hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C / 4), dim3(256), 0, s, partial, g.nslab, C, dgamma, dbeta, (const float *)nullptr, (float *)nullptr);
        // the stage-1 partials have been consumed by bn_bwd_final (stream order): the buffer is reused for the dx column sums
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(g.nslab), dim3(256), 0, s, (const T *)x, (const T *)y_relu, (const T *)dy, mean, invstd, gamma,
                           dgamma, dbeta, relu_beta, (T *)dx, M, C, g.rows, inv_m, dx_colsum ? partial : nullptr);
    } else {
        typedef float T;
        hipLaunchKernelGGL((bn_bwd_partial_kernel<T>), dim3(g.nslab), dim3(256), 0, s, (const T *)x, (const T *)y_relu, (const T *)dy, mean, invstd, gamma, relu_beta, M, C, g.rows, partial);
        hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C / 4), dim3(256), 0, s, partial, g.nslab, C, dgamma, dbeta, (const float *)nullptr, (float *)nullptr);
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(g.nslab), dim3(256), 0, s, (const T *)x, (const T *)y_relu, (const T *)dy, mean, invstd, gamma,
                           dgamma, dbeta, relu_beta, (T *)dx, M, C, g.rows, inv_m, dx_colsum ? partial : nullptr);
    }
    if (dx_colsum) hipLaunchKernelGGL(slab_sum_final_kernel, dim3(C / 4), dim3(256), 0, s, partial, g.nslab, C, dx_colsum);
    return nsg_check_launch("bn_backward");
}

/* The apply half of nsg_bn_backward alone: dgamma / dbeta are INPUTS (nsg_bn_backward_sums, or a producer that formed the
 * same sums while it wrote dy). */
int nsg_bn_backward_apply(const void *x, const void *y_relu, const void *dy, const float *mean, const float *invstd, const float *gamma,
                          const float *relu_beta, const float *dgamma, const float *dbeta, void *dx, float *dx_colsum, int64_t M, int32_t C,
                          int32_t dtype, void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && dy && mean && invstd && gamma && dx && dgamma && dbeta, NSG_E_INVALID, "nsg_bn_backward_apply: null pointer");
    int rc = check_mc("nsg_bn_backward_apply", M, C, dtype);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(x) && nsg_aligned16(dy) && nsg_aligned16(dx) && (!y_relu || nsg_aligned16(y_relu)), NSG_E_INVALID,
                "nsg_bn_backward_apply: pointers must be 16-byte aligned");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_bn_workspace_bytes(M, C), NSG_E_WORKSPACE, "nsg_bn_backward_apply: workspace too small");
    const SlabGeom g = slab_geom(M);
    hipStream_t s = (hipStream_t)stream;
    float *partial = reinterpret_cast<float *>(workspace);
    const float inv_m = 1.0f / (float)M;
    if (dtype == NSG_BF16) {
        typedef bf16_t T;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(g.nslab), dim3(256), 0, s, (const T *)x, (const T *)y_relu, (const T *)dy, mean, invstd, gamma,
                           dgamma, dbeta, relu_beta, (T *)dx, M, C, g.rows, inv_m, dx_colsum ? partial : nullptr);
    } else {
        typedef float T;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(g.nslab), dim3(256), 0, s, (const T *)x, (const T *)y_relu, (const T *)dy, mean, invstd, gamma,
                           dgamma, dbeta, relu_beta, (T *)dx, M, C, g.rows, inv_m, dx_colsum ? partial : nullptr);
    }
    if (dx_colsum) hipLaunchKernelGGL(slab_sum_final_kernel, dim3(C / 4), dim3(256), 0, s, partial, g.nslab, C, dx_colsum);
    return nsg_check_launch("bn_backward_apply");
}

/* The reduction half of nsg_bn_backward alone: dgamma, dbeta (what a fused consumer of the apply half needs first). */
int nsg_bn_backward_sums(const void *x, const void *y_relu, const void *dy, const float *mean, const float *invstd, const float *gamma,
                         const float *relu_beta, float *dgamma, float *dbeta, int64_t M, int32_t C, int32_t dtype, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && dy && mean && invstd && gamma && dgamma && dbeta, NSG_E_INVALID, "nsg_bn_backward_sums: null pointer");
    int rc = check_mc("nsg_bn_backward_sums", M, C, dtype);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(x) && nsg_aligned16(dy) && (!y_relu || nsg_aligned16(y_relu)), NSG_E_INVALID,
                "nsg_bn_backward_sums: pointers must be 16-byte aligned");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_bn_workspace_bytes(M, C), NSG_E_WORKSPACE, "nsg_bn_backward_sums: workspace too small");
    const SlabGeom g = slab_geom(M);
    hipStream_t s = (hipStream_t)stream;
    float *partial = reinterpret_cast<float *>(workspace);
    if (dtype == NSG_BF16) {
        typedef bf16_t T;
        hipLaunchKernelGGL((bn_bwd_partial_kernel<T>), dim3(g.nslab), dim3(256), 0, s, (const T *)x, (const T *)y_relu, (const T *)dy, mean, invstd, gamma, relu_beta, M, C, g.rows, partial);
    } else {
        typedef float T;
        hipLaunchKernelGGL((bn_bwd_partial_kernel<T>), dim3(g.nslab), dim3(256), 0, s, (const T *)x, (const T *)y_relu, (const T *)dy, mean, invstd, gamma, relu_beta, M, C, g.rows, partial);
    }
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C / 4), dim3(256), 0, s, partial, g.nslab, C, dgamma, dbeta, (const float *)nullptr, (float *)nullptr);
    return nsg_check_launch("bn_backward_sums");
}

}  // extern "C"
