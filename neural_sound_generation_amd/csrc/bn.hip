// BatchNorm2d over NHWC activations viewed as [M][C] (M = B*H*W).  Reference call sites:
// src/models.py:151,154,166,180 (nn.BatchNorm2d, training and eval).  HBM-bound kernels:
// float4 loads, each row of C channels is contiguous so a wave reads whole 128-B lines.
//
// Statistics are formed per slab of rows as (count, mean, M2 about the slab mean) and merged with
// Chan's parallel formula in double, in slab order: no E[x^2]-E[x]^2 cancellation and bitwise
// reproducible (no atomics).
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
    int64_t n = nsg_cdiv(M, 256);  // >= 256 rows per slab
    if (n > MAX_SLABS) n = MAX_SLABS;
    if (n < 1) n = 1;
    g.rows = (int)nsg_cdiv(M, n);
    g.nslab = (int)nsg_cdiv(M, g.rows);
    return g;
}

// threads: cg = tid % C4 owns channels 4cg..4cg+3, rg = tid / C4 strides over the slab's rows.
// partial[slab] = { mean[C], M2[C] } (count is implied by the slab geometry)
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float *__restrict__ x, int64_t M, int C, int slab_rows,
                                                               float *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    __shared__ __attribute__((aligned(16))) float smean[1024];
    const int C4 = C >> 2;
    const int rgroups = 256 / C4;
    const int tid = threadIdx.x;
    const int cg = tid % C4, rg = tid / C4;
    const bool active = rg < rgroups;
    const int64_t r0 = (int64_t)blockIdx.x * slab_rows;
    const int64_t r1 = min(M, r0 + slab_rows);
    const int n = (int)(r1 - r0);

    v4f s = {0.f, 0.f, 0.f, 0.f};
    if (active)
        for (int64_t r = r0 + rg; r < r1; r += rgroups) s += *reinterpret_cast<const v4f *>(x + r * C + cg * 4);
    if (active) *reinterpret_cast<v4f *>(red + (rg * C4 + cg) * 4) = s;
    __syncthreads();
    if (tid < C4) {
        v4f t = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < rgroups; ++g) t += *reinterpret_cast<const v4f *>(red + (g * C4 + tid) * 4);
        t = t / (float)n;
        *reinterpret_cast<v4f *>(smean + tid * 4) = t;
    }
    __syncthreads();
    v4f q = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const v4f mu = *reinterpret_cast<const v4f *>(smean + cg * 4);
        for (int64_t r = r0 + rg; r < r1; r += rgroups) {
            const v4f d = *reinterpret_cast<const v4f *>(x + r * C + cg * 4) - mu;
            q += d * d;
        }
        *reinterpret_cast<v4f *>(red + (rg * C4 + cg) * 4) = q;
    }
    __syncthreads();
    if (tid < C4) {
        v4f t = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < rgroups; ++g) t += *reinterpret_cast<const v4f *>(red + (g * C4 + tid) * 4);
        float *dst = partial + (size_t)blockIdx.x * 2 * C;
        *reinterpret_cast<v4f *>(dst + tid * 4) = *reinterpret_cast<const v4f *>(smean + tid * 4);
        *reinterpret_cast<v4f *>(dst + C + tid * 4) = t;
    }
}

// 32 lanes per channel: lane j merges its contiguous run of slabs (Chan, double), lane 0 then merges
// the 32 partial results in lane order.  Fixed order -> bitwise reproducible.
// tiled != 0: partial is [nslab][3][C] = (count, mean, M2) per row tile as written by the conv epilogue;
// tiled == 0: partial is [nslab][2][C] = (mean, M2) and the count follows from the slab geometry.
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float *__restrict__ partial, int nslab, int slab_rows, int64_t M,
                                                             int C, float eps, float momentum, float *mean, float *invstd,
                                                             float *running_mean, float *running_var, int tiled)
{
    __shared__ double sn[256], smu[256], sm2[256];
    const int tid = threadIdx.x;
    const int j = tid & 31;
    const int c = blockIdx.x * 8 + (tid >> 5);
    double n = 0.0, mu = 0.0, m2 = 0.0;
    if (c < C) {
        const int per = (nslab + 31) / 32;
        const int s0 = j * per, s1 = min(nslab, s0 + per);
        for (int s = s0; s < s1; ++s) {
            double ns, ms, qs;
            if (tiled) {
                ns = partial[(size_t)s * 3 * C + c];
                ms = partial[(size_t)s * 3 * C + C + c];
                qs = partial[(size_t)s * 3 * C + 2 * C + c];
                if (ns == 0.0) continue;
            } else {
                const int64_t r0 = (int64_t)s * slab_rows;
                ns = (double)(min(M, r0 + slab_rows) - r0);
                ms = partial[(size_t)s * 2 * C + c];
                qs = partial[(size_t)s * 2 * C + C + c];
            }
            const double tot = n + ns;
            const double delta = ms - mu;
            mu += delta * ns / tot;
            m2 += qs + delta * delta * n * ns / tot;
            n = tot;
        }
    }
    sn[tid] = n; smu[tid] = mu; sm2[tid] = m2;
    __syncthreads();
    if (j != 0 || c >= C) return;
    n = 0.0; mu = 0.0; m2 = 0.0;
    for (int k = 0; k < 32; ++k) {
        const double ns = sn[tid + k];
        if (ns == 0.0) continue;
        const double tot = n + ns;
        const double delta = smu[tid + k] - mu;
        mu += delta * ns / tot;
        m2 += sm2[tid + k] + delta * delta * n * ns / tot;
        n = tot;
    }
    const double var_b = m2 / (double)M;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var_b + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    if (running_var) {
        const double var_u = M > 1 ? m2 / (double)(M - 1) : var_b;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)var_u;
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

__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, const float *__restrict__ residual,
                                                       float *__restrict__ y, int64_t n4, int C4, int relu, int relu_res)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const v4f xv = *reinterpret_cast<const v4f *>(x + i * 4);
        const v4f mu = *reinterpret_cast<const v4f *>(mean + c);
        const v4f sc = *reinterpret_cast<const v4f *>(invstd + c) * *reinterpret_cast<const v4f *>(gamma + c);
        const v4f be = *reinterpret_cast<const v4f *>(beta + c);
        v4f o = (xv - mu) * sc + be;
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        if (residual) {
            v4f rv = *reinterpret_cast<const v4f *>(residual + i * 4);
            if (relu_res) { rv.x = fmaxf(rv.x, 0.f); rv.y = fmaxf(rv.y, 0.f); rv.z = fmaxf(rv.z, 0.f); rv.w = fmaxf(rv.w, 0.f); }
            o += rv;
        }
        *reinterpret_cast<v4f *>(y + i * 4) = o;
    }
}

// per-slab partial sums of dyh and dyh*xhat (dyh = dy masked by the fused ReLU)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float *__restrict__ x, const float *__restrict__ yrelu,
                                                             const float *__restrict__ dy, const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, int64_t M, int C,
                                                             int slab_rows, float *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float red[2 * 256 * 4];
    const int C4 = C >> 2;
    const int rgroups = 256 / C4;
    const int tid = threadIdx.x;
    const int cg = tid % C4, rg = tid / C4;
    const bool active = rg < rgroups;
    const int64_t r0 = (int64_t)blockIdx.x * slab_rows;
    const int64_t r1 = min(M, r0 + slab_rows);
    v4f s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const v4f mu = *reinterpret_cast<const v4f *>(mean + cg * 4);
        const v4f is = *reinterpret_cast<const v4f *>(invstd + cg * 4);
        for (int64_t r = r0 + rg; r < r1; r += rgroups) {
            const size_t o = (size_t)r * C + cg * 4;
            v4f g = *reinterpret_cast<const v4f *>(dy + o);
            if (yrelu) {
                const v4f yv = *reinterpret_cast<const v4f *>(yrelu + o);
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            }
            const v4f xh = (*reinterpret_cast<const v4f *>(x + o) - mu) * is;
            s1 += g;
            s2 += g * xh;
        }
        *reinterpret_cast<v4f *>(red + (rg * C4 + cg) * 4) = s1;
        *reinterpret_cast<v4f *>(red + 1024 + (rg * C4 + cg) * 4) = s2;
    }
    __syncthreads();
    if (tid < C4) {
        v4f t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < rgroups; ++g) {
            t1 += *reinterpret_cast<const v4f *>(red + (g * C4 + tid) * 4);
            t2 += *reinterpret_cast<const v4f *>(red + 1024 + (g * C4 + tid) * 4);
        }
        float *dst = partial + (size_t)blockIdx.x * 2 * C;
        *reinterpret_cast<v4f *>(dst + tid * 4) = t1;
        *reinterpret_cast<v4f *>(dst + C + tid * 4) = t2;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float *__restrict__ partial, int nslab, int C, float *dgamma, float *dbeta)
{
    __shared__ double r1[256], r2[256];
    const int tid = threadIdx.x;
    const int j = tid & 31;
    const int c = blockIdx.x * 8 + (tid >> 5);
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        const int per = (nslab + 31) / 32;
        const int b0 = j * per, b1 = min(nslab, b0 + per);
        for (int s = b0; s < b1; ++s) {
            s1 += (double)partial[(size_t)s * 2 * C + c];
            s2 += (double)partial[(size_t)s * 2 * C + C + c];
        }
    }
    r1[tid] = s1; r2[tid] = s2;
    __syncthreads();
    if (j != 0 || c >= C) return;
    s1 = 0.0; s2 = 0.0;
    for (int k = 0; k < 32; ++k) { s1 += r1[tid + k]; s2 += r2[tid + k]; }
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ yrelu,
                                                           const float *__restrict__ dy, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                           float *__restrict__ dx, int64_t n4, int C4, float inv_m)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        v4f g = *reinterpret_cast<const v4f *>(dy + i * 4);
        if (yrelu) {
            const v4f yv = *reinterpret_cast<const v4f *>(yrelu + i * 4);
            g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
            g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        }
        const v4f mu = *reinterpret_cast<const v4f *>(mean + c);
        const v4f is = *reinterpret_cast<const v4f *>(invstd + c);
        const v4f ga = *reinterpret_cast<const v4f *>(gamma + c);
        const v4f dg = *reinterpret_cast<const v4f *>(dgamma + c) * inv_m;
        const v4f db = *reinterpret_cast<const v4f *>(dbeta + c) * inv_m;
        const v4f xh = (*reinterpret_cast<const v4f *>(x + i * 4) - mu) * is;
        *reinterpret_cast<v4f *>(dx + i * 4) = (ga * is) * (g - db - xh * dg);
    }
}

// Same arithmetic, slab-structured, additionally emitting per-slab column sums of dx: dx is the
// gradient at the output of the convolution in front of this BatchNorm, so its column sum IS that
// convolution's bias gradient (saves a separate pass over dx).
__global__ __launch_bounds__(256) void bn_bwd_apply_colsum_kernel(const float *__restrict__ x, const float *__restrict__ yrelu,
                                                                  const float *__restrict__ dy, const float *__restrict__ mean,
                                                                  const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                                  const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                                  float *__restrict__ dx, int64_t M, int C, int slab_rows, float inv_m,
                                                                  float *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    const int C4 = C >> 2;
    const int rgroups = 256 / C4;
    const int tid = threadIdx.x;
    const int cg = tid % C4, rg = tid / C4;
    const bool active = rg < rgroups;
    const int64_t r0 = (int64_t)blockIdx.x * slab_rows;
    const int64_t r1 = min(M, r0 + slab_rows);
    v4f s = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const int c = cg * 4;
        const v4f mu = *reinterpret_cast<const v4f *>(mean + c);
        const v4f is = *reinterpret_cast<const v4f *>(invstd + c);
        const v4f sc = *reinterpret_cast<const v4f *>(gamma + c) * is;
        const v4f dg = *reinterpret_cast<const v4f *>(dgamma + c) * inv_m;
        const v4f db = *reinterpret_cast<const v4f *>(dbeta + c) * inv_m;
        for (int64_t r = r0 + rg; r < r1; r += rgroups) {
            const size_t o = (size_t)r * C + c;
            v4f g = *reinterpret_cast<const v4f *>(dy + o);
            if (yrelu) {
                const v4f yv = *reinterpret_cast<const v4f *>(yrelu + o);
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            }
            const v4f xh = (*reinterpret_cast<const v4f *>(x + o) - mu) * is;
            const v4f d = sc * (g - db - xh * dg);
            *reinterpret_cast<v4f *>(dx + o) = d;
            s += d;
        }
        *reinterpret_cast<v4f *>(red + (rg * C4 + cg) * 4) = s;
    }
    __syncthreads();
    if (tid < C4) {
        v4f t = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < rgroups; ++g) t += *reinterpret_cast<const v4f *>(red + (g * C4 + tid) * 4);
        *reinterpret_cast<v4f *>(partial + (size_t)blockIdx.x * C + tid * 4) = t;
    }
}

__global__ __launch_bounds__(256) void slab_sum_final_kernel(const float *__restrict__ partial, int nslab, int C, float *out)
{
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int j = tid & 31;
    const int c = blockIdx.x * 8 + (tid >> 5);
    double s = 0.0;
    if (c < C) {
        const int per = (nslab + 31) / 32;
        const int b0 = j * per, b1 = min(nslab, b0 + per);
        for (int i = b0; i < b1; ++i) s += (double)partial[(size_t)i * C + c];
    }
    red[tid] = s;
    __syncthreads();
    if (j != 0 || c >= C) return;
    s = 0.0;
    for (int k = 0; k < 32; ++k) s += red[tid + k];
    out[c] = (float)s;
}

inline int ew_blocks(int64_t n) { return (int)(nsg_cdiv(n, 256) > 4096 ? 4096 : (nsg_cdiv(n, 256) < 1 ? 1 : nsg_cdiv(n, 256))); }

inline int check_mc(const char *fn, int64_t M, int C)
{
    if (M <= 0 || C <= 0) return nsg_fail(NSG_E_INVALID, "%s: bad size", fn);
    if (C % 4 != 0 || C > 1024) return nsg_fail(NSG_E_UNSUPPORTED, "%s: C=%d must be a multiple of 4 and <= 1024", fn, C);
    if (M * (int64_t)C >= (1ll << 31)) return nsg_fail(NSG_E_UNSUPPORTED, "%s: tensor too large", fn);
    return NSG_OK;
}

}  // namespace

extern "C" {

size_t nsg_bn_workspace_bytes(int64_t M, int32_t C)
{
    if (M <= 0 || C <= 0) return 0;
    return (size_t)slab_geom(M).nslab * 2 * C * sizeof(float);
}

int nsg_bn_stats(const float *x, int64_t M, int32_t C, float eps, float momentum, float *mean, float *invstd,
                 float *running_mean, float *running_var, void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && mean && invstd, NSG_E_INVALID, "nsg_bn_stats: null pointer");
    int rc = check_mc("nsg_bn_stats", M, C);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(x), NSG_E_INVALID, "nsg_bn_stats: x must be 16-byte aligned");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_bn_workspace_bytes(M, C), NSG_E_WORKSPACE, "nsg_bn_stats: workspace too small");
    const SlabGeom g = slab_geom(M);
    hipStream_t s = (hipStream_t)stream;
    float *partial = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(g.nslab), dim3(256), 0, s, x, M, C, g.rows, partial);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + 7) / 8), dim3(256), 0, s, partial, g.nslab, g.rows, M, C, eps,
                       momentum, mean, invstd, running_mean, running_var, 0);
    return nsg_check_launch("bn_stats");
}

}  // extern "C"

namespace {
// tiles: [ntiles][3][C] = (count n_t, sum S_t, M2_t about the tile mean).  With N = sum n_t, S = sum S_t:
// mean = S/N and  M2 = sum_t ( M2_t + S_t^2/n_t ) - S^2/N  -- accumulated in double (no division per tile
// beyond S_t^2/n_t, done as a multiply by the float reciprocal the tile already rounded to).
__global__ __launch_bounds__(256) void bn_stats_tiles_final_kernel(const float *__restrict__ tiles, int ntiles, int64_t M, int C,
                                                                   float eps, float momentum, float *mean, float *invstd,
                                                                   float *running_mean, float *running_var)
{
    __shared__ double sN[256], sS[256], sQ[256];
    const int tid = threadIdx.x;
    const int j = tid & 31;
    const int c = blockIdx.x * 8 + (tid >> 5);
    double N = 0.0, S = 0.0, Q = 0.0;
    if (c < C) {
        for (int t = j; t < ntiles; t += 32) {   // interleaved: the 32 lanes of a channel stream consecutive tiles
            const double n = tiles[(size_t)t * 3 * C + c];
            if (n == 0.0) continue;
            const double st = tiles[(size_t)t * 3 * C + C + c];
            const double qt = tiles[(size_t)t * 3 * C + 2 * C + c];
            N += n;
            S += st;
            Q += qt + st * st / n;
        }
    }
    sN[tid] = N; sS[tid] = S; sQ[tid] = Q;
    __syncthreads();
    if (j != 0 || c >= C) return;
    N = 0.0; S = 0.0; Q = 0.0;
    for (int k = 0; k < 32; ++k) { N += sN[tid + k]; S += sS[tid + k]; Q += sQ[tid + k]; }
    const double mu = S / N;
    double m2 = Q - S * S / N;
    if (m2 < 0.0) m2 = 0.0;
    const double var_b = m2 / (double)M;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var_b + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    if (running_var) {
        const double var_u = M > 1 ? m2 / (double)(M - 1) : var_b;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)var_u;
    }
}
}  // namespace

// used by nsg_conv_forward_bnstats (conv_api.hip): merge the per-tile statistics the conv epilogue wrote
int nsg_bn_stats_from_tiles(const float *tiles, int ntiles, int64_t M, int C, float eps, float momentum, float *mean,
                            float *invstd, float *running_mean, float *running_var, hipStream_t s)
{
    hipLaunchKernelGGL(bn_stats_tiles_final_kernel, dim3((C + 7) / 8), dim3(256), 0, s, tiles, ntiles, M, C, eps, momentum, mean,
                       invstd, running_mean, running_var);
    return nsg_check_launch("bn_stats_from_tiles");
}

extern "C" {

int nsg_bn_eval_stats(const float *running_mean, const float *running_var, int32_t C, float eps, float *mean,
                      float *invstd, void *stream)
{
    NSG_REQUIRE(running_mean && running_var && mean && invstd && C > 0, NSG_E_INVALID, "nsg_bn_eval_stats: bad argument");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, running_mean,
                       running_var, C, eps, mean, invstd);
    return nsg_check_launch("bn_eval_stats");
}

int nsg_bn_apply(const float *x, const float *mean, const float *invstd, const float *gamma, const float *beta,
                 const float *residual, float *y, int64_t M, int32_t C, int32_t relu, int32_t relu_residual, void *stream)
{
    NSG_REQUIRE(x && mean && invstd && gamma && beta && y, NSG_E_INVALID, "nsg_bn_apply: null pointer");
    int rc = check_mc("nsg_bn_apply", M, C);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(x) && nsg_aligned16(y) && nsg_aligned16(mean) && nsg_aligned16(invstd) && nsg_aligned16(gamma) &&
                    nsg_aligned16(beta) && (!residual || nsg_aligned16(residual)),
                NSG_E_INVALID, "nsg_bn_apply: pointers must be 16-byte aligned");
    const int64_t n4 = M * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta,
                       residual, y, n4, C / 4, relu, relu_residual);
    return nsg_check_launch("bn_apply_kernel");
}

int nsg_bn_backward(const float *x, const float *y_relu, const float *dy, const float *mean, const float *invstd,
                    const float *gamma, float *dx, float *dgamma, float *dbeta, float *dx_colsum, int64_t M, int32_t C,
                    void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && dy && mean && invstd && gamma && dx && dgamma && dbeta, NSG_E_INVALID, "nsg_bn_backward: null pointer");
    int rc = check_mc("nsg_bn_backward", M, C);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(x) && nsg_aligned16(dy) && nsg_aligned16(dx) && nsg_aligned16(mean) && nsg_aligned16(invstd) &&
                    nsg_aligned16(gamma) && nsg_aligned16(dgamma) && nsg_aligned16(dbeta) && (!y_relu || nsg_aligned16(y_relu)),
                NSG_E_INVALID, "nsg_bn_backward: pointers must be 16-byte aligned");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_bn_workspace_bytes(M, C), NSG_E_WORKSPACE, "nsg_bn_backward: workspace too small");
    const SlabGeom g = slab_geom(M);
    hipStream_t s = (hipStream_t)stream;
    float *partial = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(g.nslab), dim3(256), 0, s, x, y_relu, dy, mean, invstd, M, C, g.rows, partial);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 7) / 8), dim3(256), 0, s, partial, g.nslab, C, dgamma, dbeta);
    if (dx_colsum) {
        // the stage-1 partials have been consumed by bn_bwd_final (stream order): reuse the buffer
        hipLaunchKernelGGL(bn_bwd_apply_colsum_kernel, dim3(g.nslab), dim3(256), 0, s, x, y_relu, dy, mean, invstd, gamma, dgamma,
                           dbeta, dx, M, C, g.rows, 1.0f / (float)M, partial);
        hipLaunchKernelGGL(slab_sum_final_kernel, dim3((C + 7) / 8), dim3(256), 0, s, partial, g.nslab, C, dx_colsum);
    } else {
        const int64_t n4 = M * C / 4;
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, x, y_relu, dy, mean, invstd, gamma, dgamma,
                           dbeta, dx, n4, C / 4, 1.0f / (float)M);
    }
    return nsg_check_launch("bn_backward");
}

}  // extern "C"
