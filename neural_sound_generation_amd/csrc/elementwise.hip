// Element-wise, loss and optimiser kernels (HBM-bound; float4 where alignment allows, grid-stride).
// Reference call sites: src/train.py:118-136 (zero-pad, 3x F.mse_loss, Adam step), the ReLU / Tanh
// backward of src/models.py:149,183.
#include "nsg_common.h"

namespace {

inline int ew_blocks(int64_t n)
{
    int64_t b = nsg_cdiv(n, 256);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

template <int V>
struct Vec;
template <>
struct Vec<4> { typedef v4f T; };
template <>
struct Vec<1> { typedef float T; };

// dx = (a + b) * (x > 0), elements of type T (16-byte accesses) -- nv counts 16-byte pieces
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_add_typed_kernel(const T *a, const T *b, const T *x, T *dx, int64_t nv)
{
    constexpr int W = Elem<T>::N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float g[W], xv[W];
        ldw<T, W>(a + i * W, g);
        ldw<T, W>(x + i * W, xv);
        if (b) {
            float bv[W];
            ldw<T, W>(b + i * W, bv);
#pragma unroll
            for (int e = 0; e < W; ++e) g[e] += bv[e];
        }
#pragma unroll
        for (int e = 0; e < W; ++e) g[e] = xv[e] > 0.f ? g[e] : 0.f;
        stw<T, W>(dx + i * W, g);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_add_scalar_kernel(const T *a, const T *b, const T *x, T *dx, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float g = Elem<T>::get(a + i);
        if (b) g += Elem<T>::get(b + i);
        Elem<T>::put(dx + i, Elem<T>::get(x + i) > 0.f ? g : 0.f);
    }
}

// dst = src with a change of storage type
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_kernel(const TS *src, TD *dst, int64_t n, int relu)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = Elem<TS>::get(src + i);
        Elem<TD>::put(dst + i, relu ? fmaxf(v, 0.f) : v);
    }
}
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_vec_kernel(const TS *src, TD *dst, int64_t nv, int relu)
{
    constexpr int W = Width<TS, TD>::W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float v[W];
        ldw<TS, W>(src + i * W, v);
        if (relu)
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] = fmaxf(v[e], 0.f);
        stw<TD, W>(dst + i * W, v);
    }
}

template <int V>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float *g, const float *y, float *dx, int64_t nv)
{
    typedef typename Vec<V>::T T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        const T gv = reinterpret_cast<const T *>(g)[i];
        const T yv = reinterpret_cast<const T *>(y)[i];
        reinterpret_cast<T *>(dx)[i] = gv * (1.0f - yv * yv);
    }
}

template <int V>
__global__ __launch_bounds__(256) void add_kernel(const float *a, const float *b, float *y, int64_t nv)
{
    typedef typename Vec<V>::T T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        T v = reinterpret_cast<const T *>(a)[i];
        if (b) v += reinterpret_cast<const T *>(b)[i];
        reinterpret_cast<T *>(y)[i] = v;
    }
}

// d loss_vq / d codebook from the per-code statistics of the encoder rows: out[k][d] = scale * (n[k] * e[k][d] - s[k][d])
__global__ __launch_bounds__(256) void codebook_grad_from_sums_kernel(const float *__restrict__ e, const float *__restrict__ n,
                                                                      const float *__restrict__ s, float *__restrict__ out, int64_t total,
                                                                      int D, float scale)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i / D;
        const float ne = e[i] * n[k];          // three separately rounded operations (-ffp-contract=off), as ATen's mul, sub_, mul_ did
        out[i] = (ne - s[i]) * scale;
    }
}

struct CounterPtrs {
    int64_t *p[32];
};
__global__ void increment_counters_kernel(const CounterPtrs c, int n)
{
    const int i = threadIdx.x;
    if (i < n) *c.p[i] += 1;
}

// block-level sum in double, fixed order: thread partials -> LDS -> thread 0 walks them
__device__ __forceinline__ void block_sum_store(double part, double *dst)
{
    __shared__ double red[256];
    red[threadIdx.x] = part;
    __syncthreads();
    if (threadIdx.x < 64) {
        double t = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
        red[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 64; ++i) t += red[i];
        dst[blockIdx.x] = t;
    }
}

// one wave: each lane sums a strided share of the block partials, then a fixed-order lane walk
__global__ void final_mean_kernel(const double *partial, int n, double denom, float *out)
{
    __shared__ double red[64];
    const int lane = threadIdx.x;
    double t = 0.0;
#pragma unroll 8
    for (int i = lane; i < n; i += 64) t += partial[i];
    red[lane] = t;
    __syncthreads();
    if (lane == 0) {
        double s = 0.0;
        for (int i = 0; i < 64; ++i) s += red[i];
        out[0] = (float)(s / denom);
    }
}

// sum over [rows][wc] of (pad(a) - c)^2, a is [rows][wa] (wa <= wc); optional gradient wrt a
__global__ __launch_bounds__(256) void mse_padded_kernel(const float *__restrict__ a, const float *__restrict__ c, int64_t rows,
                                                         int wa, int wc, float gscale, float *__restrict__ da, double *partial)
{
    const int64_t n = rows * wc;
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / wc;
        const int col = (int)(i - r * wc);
        const float cv = c[i];
        float d;
        if (col < wa) {
            d = a[r * wa + col] - cv;
            if (da) da[r * wa + col] = gscale * d;
        } else {
            d = -cv;
        }
        acc += (double)(d * d);
    }
    block_sum_store(acc, partial);
}

// z, q fp32 (the VQ works in fp32 in both modes); dz / dz_add of type TG (the encoder-side gradient),
// dq fp32 (it feeds the fp32 codebook scatter).  n counts elements; VEC: 8 elements per thread.
template <typename TG, bool VEC>
__global__ __launch_bounds__(256) void vq_losses_kernel(const float *__restrict__ z, const float *__restrict__ q, int64_t n,
                                                        float zscale, float qscale, const TG *__restrict__ dz_add,
                                                        TG *__restrict__ dz, float *__restrict__ dq, double *partial)
{
    constexpr int W = VEC ? 8 : 1;
    const int64_t nv = n / W;
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float zv[W], qv[W], d[W];
        if constexpr (VEC) { ldw<float, 8>(z + i * 8, zv); ldw<float, 8>(q + i * 8, qv); }
        else { zv[0] = z[i]; qv[0] = q[i]; }
#pragma unroll
        for (int e = 0; e < W; ++e) { d[e] = zv[e] - qv[e]; acc += (double)(d[e] * d[e]); }
        if (dz) {
            float g[W];
#pragma unroll
            for (int e = 0; e < W; ++e) g[e] = d[e] * zscale;
            if (dz_add) {
                float av[W];
                if constexpr (VEC) ldw<TG, 8>(dz_add + i * 8, av); else av[0] = Elem<TG>::get(dz_add + i);
#pragma unroll
                for (int e = 0; e < W; ++e) g[e] += av[e];
            }
            if constexpr (VEC) stw<TG, 8>(dz + i * 8, g); else Elem<TG>::put(dz + i, g[0]);
        }
        if (dq) {
            float g[W];
#pragma unroll
            for (int e = 0; e < W; ++e) g[e] = d[e] * (-qscale);
            if constexpr (VEC) stw<float, 8>(dq + i * 8, g); else dq[i] = g[0];
        }
    }
    block_sum_store(acc, partial);
}

// The same loss and encoder-side gradient with q given as (codebook, indices): q[row] = e[idx[row]] is read from the
// L2-resident codebook instead of a materialised [N][D] tensor.  D % 8 == 0; a thread takes 8 channels of one row.
template <typename T> __device__ __forceinline__ float round_as(float v);       // v as it reads back after a store as T
template <> __device__ __forceinline__ float round_as<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_as<bf16_t>(float v) { return nsg_bf2f(nsg_f2bf(v)); }

// Slab-structured exactly as bn_bwd_partial_kernel (bn.hip): block = one slab of slab_rows rows, thread (cg = tid % (D / 8),
// rg = tid / (D / 8)) takes channels 8 cg .. + 7 of rows r0 + rg, + 256 / (D / 8), ...  (threads past the last whole row group idle).
// BN: dz is the incoming gradient of a BatchNorm whose input is bn_x (the encoder's last ResBlock, src/models.py:154): the two
// sums of that BatchNorm's backward -- bn_partial[slab][2][D] = (sum dz, sum dz * xhat) of the values AS STORED -- are formed
// while dz is written, with bn_bwd_partial_kernel's expressions in its order: for bf16 gradients (8 channels per thread there
// too) the finished sums equal those of the separate pass over (bn_x, dz) bit for bit.
template <typename TG, bool BN>
__global__ __launch_bounds__(256) void vq_losses_indexed_kernel(const float *__restrict__ z, const float *__restrict__ e,
                                                                const int64_t *__restrict__ idx, int64_t N, int D, int K, float zscale,
                                                                const TG *__restrict__ dz_add, TG *__restrict__ dz, double *partial,
                                                                int slab_rows, const TG *__restrict__ bn_x,
                                                                const float *__restrict__ bn_mean, const float *__restrict__ bn_invstd,
                                                                float *__restrict__ bn_partial)
{
    const int D8 = D >> 3;
    const int rgroups = 256 / D8;
    const int cg = threadIdx.x % D8, rg = threadIdx.x / D8;
    const int d8 = cg * 8;
    const int64_t r0 = (int64_t)blockIdx.x * slab_rows;
    const int64_t r1 = min(N, r0 + (int64_t)slab_rows);
    double acc = 0.0;
    float s1[8], s2[8], mu[8], is[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { s1[c] = 0.f; s2[c] = 0.f; mu[c] = BN ? bn_mean[d8 + c] : 0.f; is[c] = BN ? bn_invstd[d8 + c] : 0.f; }
    for (int64_t row = r0 + rg; row < r1 && rg < rgroups; row += rgroups) {
        const int64_t i = row * D8 + cg;
        int64_t k = idx[row];
        k = k < 0 ? 0 : (k >= K ? K - 1 : k);                 // (validated indices; clamped so a bad one cannot fault)
        float zv[8], qv[8], g[8];
        ldw<float, 8>(z + i * 8, zv);
        ldw<float, 8>(e + (size_t)k * D + d8, qv);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float d = zv[c] - qv[c];
            acc += (double)(d * d);
            g[c] = d * zscale;
        }
        if (dz) {
            if (dz_add) {
                float av[8];
                ldw<TG, 8>(dz_add + i * 8, av);
#pragma unroll
                for (int c = 0; c < 8; ++c) g[c] += av[c];
            }
            stw<TG, 8>(dz + i * 8, g);
            if constexpr (BN) {
                float xv[8];
                ldw<TG, 8>(bn_x + i * 8, xv);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float gs = round_as<TG>(g[c]);
                    s1[c] += gs;
                    s2[c] += gs * ((xv[c] - mu[c]) * is[c]);
                }
            }
        }
    }
    block_sum_store(acc, partial);
    if constexpr (BN) {
        __shared__ float red[2 * 256 * 8];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (rg < rgroups) { red[(rg * D8 + cg) * 8 + c] = s1[c]; red[256 * 8 + (rg * D8 + cg) * 8 + c] = s2[c]; }
        __syncthreads();
        if ((int)threadIdx.x < D8) {
            float *dst = bn_partial + (size_t)blockIdx.x * 2 * D;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float t1 = 0.f, t2 = 0.f;
                for (int gq = 0; gq < rgroups; ++gq) { t1 += red[(gq * D8 + threadIdx.x) * 8 + c]; t2 += red[256 * 8 + (gq * D8 + threadIdx.x) * 8 + c]; }
                dst[threadIdx.x * 8 + c] = t1;
                dst[D + threadIdx.x * 8 + c] = t2;
            }
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void adam_kernel(float *p, const float *g, float *m, float *v, int64_t nv, float b1, float b2,
                                                   float eps, float step_size, float bc2_sqrt, float gscale)
{
    typedef typename Vec<V>::T T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        const T gv = reinterpret_cast<const T *>(g)[i] * gscale;
        T mv = reinterpret_cast<T *>(m)[i];
        T vv = reinterpret_cast<T *>(v)[i];
        mv = mv + (gv - mv) * (1.0f - b1);          // lerp, as torch's _single_tensor_adam
        vv = vv * b2 + gv * gv * (1.0f - b2);
        T den;
        if constexpr (V == 4) {
            den.x = sqrtf(vv.x) / bc2_sqrt + eps; den.y = sqrtf(vv.y) / bc2_sqrt + eps;
            den.z = sqrtf(vv.z) / bc2_sqrt + eps; den.w = sqrtf(vv.w) / bc2_sqrt + eps;
        } else {
            den = sqrtf(vv) / bc2_sqrt + eps;
        }
        reinterpret_cast<T *>(m)[i] = mv;
        reinterpret_cast<T *>(v)[i] = vv;
        reinterpret_cast<T *>(p)[i] = reinterpret_cast<T *>(p)[i] - (mv / den) * step_size;
    }
}

// y[b][r][:] = x[b][r][:] + rows[b][:]   (speaker embedding broadcast over a clip's pixels); x, rows fp32, y of type TY
template <typename TY>
__global__ __launch_bounds__(256) void add_per_clip_kernel(const float *__restrict__ x, const float *__restrict__ rows, TY *__restrict__ y,
                                                           int64_t nw, int CW, int64_t per_clip_w)
{
    constexpr int W = Width<float, TY>::W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nw; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per_clip_w;
        const int c = (int)(i % CW);
        float xv[W], rv[W];
        ldw<float, W>(x + i * W, xv);
        ldw<float, W>(rows + (b * CW + c) * W, rv);
#pragma unroll
        for (int e = 0; e < W; ++e) xv[e] += rv[e];
        stw<TY, W>(y + i * W, xv);
    }
}

// partial[b][s][:] = sum of rows [s*R, (s+1)*R) of clip b;  then out[b][:] = sum_s partial (fixed order)
template <typename T>
__global__ __launch_bounds__(256) void clip_colsum_partial_kernel(const T *__restrict__ x, int rows_per_clip, int C, int slabs,
                                                                  float *__restrict__ partial)
{
    constexpr int W = Elem<T>::N;
    __shared__ float red[256 * W];
    const int CW = C / W;
    const int rgroups = 256 / CW;
    const int tid = threadIdx.x;
    const int cg = tid % CW, rg = tid / CW;
    const int b = blockIdx.x / slabs, sl = blockIdx.x % slabs;
    const int R = (rows_per_clip + slabs - 1) / slabs;
    const int r0 = sl * R, r1 = min(rows_per_clip, r0 + R);
    const T *base = x + (size_t)b * rows_per_clip * C;
    if (rg < rgroups) {
        float s[W];
#pragma unroll
        for (int e = 0; e < W; ++e) s[e] = 0.f;
        for (int r = r0 + rg; r < r1; r += rgroups) {
            float v[W];
            ldw<T, W>(base + (size_t)r * C + cg * W, v);
#pragma unroll
            for (int e = 0; e < W; ++e) s[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < W; ++e) red[(rg * CW + cg) * W + e] = s[e];
    }
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
__global__ void clip_colsum_final_kernel(const float *__restrict__ partial, int B, int slabs, int C, float *__restrict__ out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * C) return;
    const int b = (int)(i / C), c = (int)(i % C);
    float s = 0.f;
    for (int k = 0; k < slabs; ++k) s += partial[((size_t)b * slabs + k) * C + c];
    out[i] = s;
}

constexpr int CLIP_SLABS = 16;
constexpr int RED_BLOCKS = 1024;

}  // namespace

// out[0] = (sum of n block partials) / denom: the closing kernel of every loss (also conv_api.hip's fused output layer + loss)
int nsg_launch_final_mean(const double *partial, int n, double denom, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(final_mean_kernel, dim3(1), dim3(64), 0, s, partial, n, denom, out);
    return nsg_check_launch("final_mean_kernel");
}

extern "C" {

int nsg_relu_backward_add(const void *a, const void *b, const void *x, void *dx, int64_t n, int32_t dtype, void *stream)
{
    NSG_REQUIRE(a && x && dx && n >= 0, NSG_E_INVALID, "nsg_relu_backward_add: bad argument");
    NSG_REQUIRE(dtype == NSG_F32 || dtype == NSG_BF16, NSG_E_INVALID, "nsg_relu_backward_add: unknown dtype");
    if (n == 0) return NSG_OK;
    hipStream_t s = (hipStream_t)stream;
    const int W = dtype == NSG_BF16 ? 8 : 4;
    const bool vec = (n % W) == 0 && nsg_aligned16(a) && nsg_aligned16(x) && nsg_aligned16(dx) && (!b || nsg_aligned16(b));
    if (dtype == NSG_BF16) {
        typedef bf16_t T;
        if (vec) hipLaunchKernelGGL((relu_bwd_add_typed_kernel<T>), dim3(ew_blocks(n / W)), dim3(256), 0, s, (const T *)a, (const T *)b, (const T *)x, (T *)dx, n / W);
        else     hipLaunchKernelGGL((relu_bwd_add_scalar_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, s, (const T *)a, (const T *)b, (const T *)x, (T *)dx, n);
    } else {
        typedef float T;
        if (vec) hipLaunchKernelGGL((relu_bwd_add_typed_kernel<T>), dim3(ew_blocks(n / W)), dim3(256), 0, s, (const T *)a, (const T *)b, (const T *)x, (T *)dx, n / W);
        else     hipLaunchKernelGGL((relu_bwd_add_scalar_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, s, (const T *)a, (const T *)b, (const T *)x, (T *)dx, n);
    }
    return nsg_check_launch("relu_bwd_add_kernel");
}

int nsg_convert(const void *src, int32_t src_dtype, void *dst, int32_t dst_dtype, int64_t n, int32_t relu, void *stream)
{
    NSG_REQUIRE(src && dst && n >= 0, NSG_E_INVALID, "nsg_convert: bad argument");
    NSG_REQUIRE((src_dtype == NSG_F32 || src_dtype == NSG_BF16) && (dst_dtype == NSG_F32 || dst_dtype == NSG_BF16), NSG_E_INVALID,
                "nsg_convert: unknown dtype");
    if (n == 0) return NSG_OK;
    hipStream_t s = (hipStream_t)stream;
    const int W = (src_dtype == NSG_BF16 || dst_dtype == NSG_BF16) ? 8 : 4;
    const bool vec = (n % W) == 0 && nsg_aligned16(src) && nsg_aligned16(dst);
#define NSG_CVT(TS, TD)                                                                                                                   \
    do {                                                                                                                                  \
        if (vec) hipLaunchKernelGGL((convert_vec_kernel<TS, TD>), dim3(ew_blocks(n / W)), dim3(256), 0, s, (const TS *)src, (TD *)dst, n / W, relu); \
        else     hipLaunchKernelGGL((convert_kernel<TS, TD>), dim3(ew_blocks(n)), dim3(256), 0, s, (const TS *)src, (TD *)dst, n, relu);          \
    } while (0)
    if (src_dtype == NSG_F32 && dst_dtype == NSG_BF16) NSG_CVT(float, bf16_t);
    else if (src_dtype == NSG_BF16 && dst_dtype == NSG_F32) NSG_CVT(bf16_t, float);
    else if (src_dtype == NSG_F32) NSG_CVT(float, float);
    else NSG_CVT(bf16_t, bf16_t);
#undef NSG_CVT
    return nsg_check_launch("convert_kernel");
}

int nsg_tanh_backward(const float *g, const float *y, float *dx, int64_t n, void *stream)
{
    NSG_REQUIRE(g && y && dx && n >= 0, NSG_E_INVALID, "nsg_tanh_backward: bad argument");
    if (n == 0) return NSG_OK;
    hipStream_t s = (hipStream_t)stream;
    if ((n & 3) == 0 && nsg_aligned16(g) && nsg_aligned16(y) && nsg_aligned16(dx))
        hipLaunchKernelGGL((tanh_bwd_kernel<4>), dim3(ew_blocks(n / 4)), dim3(256), 0, s, g, y, dx, n / 4);
    else
        hipLaunchKernelGGL((tanh_bwd_kernel<1>), dim3(ew_blocks(n)), dim3(256), 0, s, g, y, dx, n);
    return nsg_check_launch("tanh_bwd_kernel");
}

int nsg_add(const float *a, const float *b, float *y, int64_t n, void *stream)
{
    NSG_REQUIRE(a && y && n >= 0, NSG_E_INVALID, "nsg_add: bad argument");
    if (n == 0) return NSG_OK;
    hipStream_t s = (hipStream_t)stream;
    if ((n & 3) == 0 && nsg_aligned16(a) && nsg_aligned16(y) && (!b || nsg_aligned16(b)))
        hipLaunchKernelGGL((add_kernel<4>), dim3(ew_blocks(n / 4)), dim3(256), 0, s, a, b, y, n / 4);
    else
        hipLaunchKernelGGL((add_kernel<1>), dim3(ew_blocks(n)), dim3(256), 0, s, a, b, y, n);
    return nsg_check_launch("add_kernel");
}

int nsg_codebook_grad_from_sums(const float *e, const float *n, const float *s, int32_t K, int32_t D, float scale, float *out, void *stream)
{
    NSG_REQUIRE(e && n && s && out && K > 0 && D > 0, NSG_E_INVALID, "nsg_codebook_grad_from_sums: bad argument");
    const int64_t total = (int64_t)K * D;
    hipLaunchKernelGGL(codebook_grad_from_sums_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, e, n, s, out, total, (int)D, scale);
    return nsg_check_launch("codebook_grad_from_sums_kernel");
}

int nsg_increment_counters(int64_t *const *counters, int32_t n, void *stream)
{
    NSG_REQUIRE(n >= 0 && (n == 0 || counters), NSG_E_INVALID, "nsg_increment_counters: bad argument");
    for (int32_t i0 = 0; i0 < n; i0 += 32) {
        CounterPtrs c = {};
        const int m = n - i0 < 32 ? n - i0 : 32;
        for (int i = 0; i < m; ++i) {
            NSG_REQUIRE(counters[i0 + i], NSG_E_INVALID, "nsg_increment_counters: null counter");
            c.p[i] = counters[i0 + i];
        }
        hipLaunchKernelGGL(increment_counters_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c, m);
        const int rc = nsg_check_launch("increment_counters_kernel");
        if (rc) return rc;
    }
    return NSG_OK;
}

int nsg_add_per_clip(const float *x, const float *rows, void *y, int32_t B, int64_t rows_per_clip, int32_t C, int32_t y_dtype,
                     void *stream)
{
    NSG_REQUIRE(x && rows && y && B > 0 && rows_per_clip > 0 && C > 0, NSG_E_INVALID, "nsg_add_per_clip: bad argument");
    NSG_REQUIRE(y_dtype == NSG_F32 || y_dtype == NSG_BF16, NSG_E_INVALID, "nsg_add_per_clip: unknown dtype");
    const int W = y_dtype == NSG_BF16 ? 8 : 4;
    NSG_REQUIRE(C % W == 0 && nsg_aligned16(x) && nsg_aligned16(rows) && nsg_aligned16(y), NSG_E_UNSUPPORTED,
                "nsg_add_per_clip: C must be a multiple of %d and pointers 16-byte aligned", W);
    const int64_t nw = (int64_t)B * rows_per_clip * C / W;
    if (y_dtype == NSG_BF16)
        hipLaunchKernelGGL((add_per_clip_kernel<bf16_t>), dim3(ew_blocks(nw)), dim3(256), 0, (hipStream_t)stream, x, rows, (bf16_t *)y, nw, C / W, rows_per_clip * (C / W));
    else
        hipLaunchKernelGGL((add_per_clip_kernel<float>), dim3(ew_blocks(nw)), dim3(256), 0, (hipStream_t)stream, x, rows, (float *)y, nw, C / W, rows_per_clip * (C / W));
    return nsg_check_launch("add_per_clip_kernel");
}

size_t nsg_clip_colsum_workspace_bytes(int32_t B, int32_t C) { return (size_t)(B > 0 ? B : 0) * CLIP_SLABS * (C > 0 ? C : 0) * sizeof(float); }

int nsg_clip_colsum(const void *x, int32_t dtype, int32_t B, int64_t rows_per_clip, int32_t C, float *out, void *workspace,
                    size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && out && B > 0 && rows_per_clip > 0 && C > 0, NSG_E_INVALID, "nsg_clip_colsum: bad argument");
    NSG_REQUIRE(dtype == NSG_F32 || dtype == NSG_BF16, NSG_E_INVALID, "nsg_clip_colsum: unknown dtype");
    const int W = dtype == NSG_BF16 ? 8 : 4;
    NSG_REQUIRE(C % W == 0 && C <= 1024 && nsg_aligned16(x) && rows_per_clip < (1ll << 31), NSG_E_UNSUPPORTED,
                "nsg_clip_colsum: C must be a multiple of %d (<= 1024), x 16-byte aligned", W);
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_clip_colsum_workspace_bytes(B, C), NSG_E_WORKSPACE, "nsg_clip_colsum: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float *partial = reinterpret_cast<float *>(workspace);
    if (dtype == NSG_BF16)
        hipLaunchKernelGGL((clip_colsum_partial_kernel<bf16_t>), dim3(B * CLIP_SLABS), dim3(256), 0, s, (const bf16_t *)x, (int)rows_per_clip, C, CLIP_SLABS, partial);
    else
        hipLaunchKernelGGL((clip_colsum_partial_kernel<float>), dim3(B * CLIP_SLABS), dim3(256), 0, s, (const float *)x, (int)rows_per_clip, C, CLIP_SLABS, partial);
    hipLaunchKernelGGL(clip_colsum_final_kernel, dim3((unsigned)nsg_cdiv((int64_t)B * C, 256)), dim3(256), 0, s, partial, B, CLIP_SLABS, C, out);
    return nsg_check_launch("clip_colsum");
}

size_t nsg_reduce_workspace_bytes(int64_t n)
{
    (void)n;
    return (size_t)RED_BLOCKS * sizeof(double);
}

int nsg_mse_padded(const float *a, const float *c, int64_t rows, int32_t wa, int32_t wc, float grad_scale, float *loss_out,
                   float *da, void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(a && c && loss_out && rows > 0 && wa > 0 && wc >= wa, NSG_E_INVALID, "nsg_mse_padded: bad argument");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_reduce_workspace_bytes(rows * wc), NSG_E_WORKSPACE, "nsg_mse_padded: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = rows * (int64_t)wc;
    int nb = ew_blocks(n);
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    double *partial = reinterpret_cast<double *>(workspace);
    hipLaunchKernelGGL(mse_padded_kernel, dim3(nb), dim3(256), 0, s, a, c, rows, wa, wc, grad_scale * 2.0f / (float)n, da, partial);
    hipLaunchKernelGGL(final_mean_kernel, dim3(1), dim3(64), 0, s, partial, nb, (double)n, loss_out);
    return nsg_check_launch("mse_padded");
}

int nsg_vq_losses(const float *z, const float *q, int64_t n, float dz_scale, float dq_scale, const void *dz_add,
                  float *loss_out, void *dz, float *dq, int32_t grad_dtype, void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(z && q && loss_out && n > 0, NSG_E_INVALID, "nsg_vq_losses: bad argument");
    NSG_REQUIRE(grad_dtype == NSG_F32 || grad_dtype == NSG_BF16, NSG_E_INVALID, "nsg_vq_losses: unknown dtype");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_reduce_workspace_bytes(n), NSG_E_WORKSPACE, "nsg_vq_losses: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    double *partial = reinterpret_cast<double *>(workspace);
    const float zs = dz_scale * 2.0f / (float)n, qs = dq_scale * 2.0f / (float)n;
    const bool vec = (n & 7) == 0 && nsg_aligned16(z) && nsg_aligned16(q) && (!dz || nsg_aligned16(dz)) &&
                     (!dq || nsg_aligned16(dq)) && (!dz_add || nsg_aligned16(dz_add));
    int nb = ew_blocks(vec ? n / 8 : n);
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
#define NSG_VQL(TG, V) hipLaunchKernelGGL((vq_losses_kernel<TG, V>), dim3(nb), dim3(256), 0, s, z, q, n, zs, qs, (const TG *)dz_add, (TG *)dz, dq, partial)
    if (grad_dtype == NSG_BF16) { if (vec) NSG_VQL(bf16_t, true); else NSG_VQL(bf16_t, false); }
    else                        { if (vec) NSG_VQL(float, true);  else NSG_VQL(float, false); }
#undef NSG_VQL
    hipLaunchKernelGGL(final_mean_kernel, dim3(1), dim3(64), 0, s, partial, nb, (double)n, loss_out);
    return nsg_check_launch("vq_losses");
}

int nsg_vq_losses_indexed(const float *z, const float *codebook, const int64_t *idx, int64_t N, int32_t D, int32_t K, float dz_scale,
                          const void *dz_add, float *loss_out, void *dz, int32_t grad_dtype, void *workspace, size_t workspace_bytes,
                          void *stream)
{
    NSG_REQUIRE(z && codebook && idx && loss_out && N > 0 && D > 0 && K > 0, NSG_E_INVALID, "nsg_vq_losses_indexed: bad argument");
    NSG_REQUIRE(D % 8 == 0 && D <= 2048, NSG_E_UNSUPPORTED, "nsg_vq_losses_indexed: D=%d must be a multiple of 8 up to 2048", D);
    NSG_REQUIRE(grad_dtype == NSG_F32 || grad_dtype == NSG_BF16, NSG_E_INVALID, "nsg_vq_losses_indexed: grad_dtype must be NSG_F32 or NSG_BF16");
    NSG_REQUIRE(nsg_aligned16(z) && nsg_aligned16(codebook) && (!dz || nsg_aligned16(dz)) && (!dz_add || nsg_aligned16(dz_add)), NSG_E_INVALID,
                "nsg_vq_losses_indexed: pointers must be 16-byte aligned");
    const int64_t n = N * D;
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_reduce_workspace_bytes(n), NSG_E_WORKSPACE, "nsg_vq_losses_indexed: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    double *partial = reinterpret_cast<double *>(workspace);
    const float zs = dz_scale * 2.0f / (float)n;
    int nb, rows;
    nsg_bn_slab_geom(N, &nb, &rows);
    static_assert(RED_BLOCKS >= 1024, "one loss partial per slab");
    if (grad_dtype == NSG_BF16)
        hipLaunchKernelGGL((vq_losses_indexed_kernel<bf16_t, false>), dim3(nb), dim3(256), 0, s, z, codebook, idx, N, D, K, zs, (const bf16_t *)dz_add, (bf16_t *)dz, partial,
                           rows, (const bf16_t *)nullptr, (const float *)nullptr, (const float *)nullptr, (float *)nullptr);
    else
        hipLaunchKernelGGL((vq_losses_indexed_kernel<float, false>), dim3(nb), dim3(256), 0, s, z, codebook, idx, N, D, K, zs, (const float *)dz_add, (float *)dz, partial,
                           rows, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (float *)nullptr);
    hipLaunchKernelGGL(final_mean_kernel, dim3(1), dim3(64), 0, s, partial, nb, (double)n, loss_out);
    return nsg_check_launch("vq_losses_indexed");
}

int32_t nsg_vq_losses_indexed_bn_supported(int32_t D) { return D >= 8 && D % 8 == 0 && D <= 2048 ? 1 : 0; }

size_t nsg_vq_losses_indexed_bn_workspace_bytes(int64_t N, int32_t D)
{
    if (N <= 0 || D <= 0) return 0;
    return nsg_align_up(nsg_reduce_workspace_bytes(N * D), 256) + (size_t)RED_BLOCKS * 2 * D * sizeof(float);
}

int nsg_vq_losses_indexed_bn(const float *z, const float *codebook, const int64_t *idx, int64_t N, int32_t D, int32_t K, float dz_scale,
                             const void *dz_add, float *loss_out, void *dz, int32_t grad_dtype, const void *bn_x, const float *bn_mean,
                             const float *bn_invstd, float *bn_dgamma, float *bn_dbeta, void *workspace, size_t workspace_bytes,
                             void *stream)
{
    NSG_REQUIRE(z && codebook && idx && loss_out && dz && bn_x && bn_mean && bn_invstd && bn_dgamma && bn_dbeta && N > 0 && K > 0, NSG_E_INVALID,
                "nsg_vq_losses_indexed_bn: bad argument");
    NSG_REQUIRE(nsg_vq_losses_indexed_bn_supported(D), NSG_E_UNSUPPORTED, "nsg_vq_losses_indexed_bn: D=%d must be a multiple of 8 up to 2048", D);
    NSG_REQUIRE(grad_dtype == NSG_F32 || grad_dtype == NSG_BF16, NSG_E_INVALID, "nsg_vq_losses_indexed_bn: grad_dtype must be NSG_F32 or NSG_BF16");
    NSG_REQUIRE(nsg_aligned16(z) && nsg_aligned16(codebook) && nsg_aligned16(dz) && nsg_aligned16(bn_x) && (!dz_add || nsg_aligned16(dz_add)), NSG_E_INVALID,
                "nsg_vq_losses_indexed_bn: pointers must be 16-byte aligned");
    const int64_t n = N * D;
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_vq_losses_indexed_bn_workspace_bytes(N, D), NSG_E_WORKSPACE, "nsg_vq_losses_indexed_bn: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    double *partial = reinterpret_cast<double *>(workspace);
    float *bnp = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + nsg_align_up(nsg_reduce_workspace_bytes(n), 256));
    const float zs = dz_scale * 2.0f / (float)n;
    int nb, rows;
    nsg_bn_slab_geom(N, &nb, &rows);
    if (grad_dtype == NSG_BF16)
        hipLaunchKernelGGL((vq_losses_indexed_kernel<bf16_t, true>), dim3(nb), dim3(256), 0, s, z, codebook, idx, N, D, K, zs, (const bf16_t *)dz_add, (bf16_t *)dz, partial,
                           rows, (const bf16_t *)bn_x, bn_mean, bn_invstd, bnp);
    else
        hipLaunchKernelGGL((vq_losses_indexed_kernel<float, true>), dim3(nb), dim3(256), 0, s, z, codebook, idx, N, D, K, zs, (const float *)dz_add, (float *)dz, partial,
                           rows, (const float *)bn_x, bn_mean, bn_invstd, bnp);
    hipLaunchKernelGGL(final_mean_kernel, dim3(1), dim3(64), 0, s, partial, nb, (double)n, loss_out);
    int rc = nsg_check_launch("vq_losses_indexed_bn");
    if (rc) return rc;
    return nsg_launch_bn_bwd_final(bnp, nb, D, bn_dgamma, bn_dbeta, s);
}

int nsg_adam_step(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2, float eps,
                  int32_t step, float grad_scale, void *stream)
{
    NSG_REQUIRE(p && g && m && v && n >= 0 && step >= 1, NSG_E_INVALID, "nsg_adam_step: bad argument");
    if (n == 0) return NSG_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipStream_t s = (hipStream_t)stream;
    if ((n & 3) == 0 && nsg_aligned16(p) && nsg_aligned16(g) && nsg_aligned16(m) && nsg_aligned16(v))
        hipLaunchKernelGGL((adam_kernel<4>), dim3(ew_blocks(n / 4)), dim3(256), 0, s, p, g, m, v, n / 4, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale);
    else
        hipLaunchKernelGGL((adam_kernel<1>), dim3(ew_blocks(n)), dim3(256), 0, s, p, g, m, v, n, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale);
    return nsg_check_launch("adam_kernel");
}

}  // extern "C"
