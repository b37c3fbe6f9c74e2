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

template <int V>
__global__ __launch_bounds__(256) void relu_bwd_add_kernel(const float *a, const float *b, const float *x, float *dx, int64_t nv)
{
    typedef typename Vec<V>::T T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        T g = reinterpret_cast<const T *>(a)[i];
        if (b) g += reinterpret_cast<const T *>(b)[i];
        const T xv = reinterpret_cast<const T *>(x)[i];
        if constexpr (V == 4) {
            g.x = xv.x > 0.f ? g.x : 0.f; g.y = xv.y > 0.f ? g.y : 0.f;
            g.z = xv.z > 0.f ? g.z : 0.f; g.w = xv.w > 0.f ? g.w : 0.f;
        } else {
            g = xv > 0.f ? g : 0.f;
        }
        reinterpret_cast<T *>(dx)[i] = g;
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

__global__ void final_mean_kernel(const double *partial, int n, double denom, float *out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n; ++i) t += partial[i];
        out[0] = (float)(t / denom);
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

template <int V>
__global__ __launch_bounds__(256) void vq_losses_kernel(const float *__restrict__ z, const float *__restrict__ q, int64_t nv,
                                                        float zscale, float qscale, const float *__restrict__ dz_add,
                                                        float *__restrict__ dz, float *__restrict__ dq, double *partial)
{
    typedef typename Vec<V>::T T;
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        const T d = reinterpret_cast<const T *>(z)[i] - reinterpret_cast<const T *>(q)[i];
        if constexpr (V == 4) {
            acc += (double)(d.x * d.x) + (double)(d.y * d.y) + (double)(d.z * d.z) + (double)(d.w * d.w);
        } else {
            acc += (double)(d * d);
        }
        if (dz) {
            T g = d * zscale;
            if (dz_add) g += reinterpret_cast<const T *>(dz_add)[i];
            reinterpret_cast<T *>(dz)[i] = g;
        }
        if (dq) reinterpret_cast<T *>(dq)[i] = d * (-qscale);
    }
    block_sum_store(acc, partial);
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

// y[b][r][:] = x[b][r][:] + rows[b][:]   (speaker embedding broadcast over a clip's pixels)
__global__ __launch_bounds__(256) void add_per_clip_kernel(const float *__restrict__ x, const float *__restrict__ rows, float *__restrict__ y,
                                                           int64_t n4, int C4, int64_t per_clip4)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per_clip4;
        const int c = (int)(i % C4);
        const v4f r = *reinterpret_cast<const v4f *>(rows + (b * C4 + c) * 4);
        *reinterpret_cast<v4f *>(y + i * 4) = *reinterpret_cast<const v4f *>(x + i * 4) + r;
    }
}

// partial[b][s][:] = sum of rows [s*R, (s+1)*R) of clip b;  then out[b][:] = sum_s partial (fixed order)
__global__ __launch_bounds__(256) void clip_colsum_partial_kernel(const float *__restrict__ x, int rows_per_clip, int C, int slabs,
                                                                  float *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    const int C4 = C >> 2;
    const int rgroups = 256 / C4;
    const int tid = threadIdx.x;
    const int cg = tid % C4, rg = tid / C4;
    const int b = blockIdx.x / slabs, sl = blockIdx.x % slabs;
    const int R = (rows_per_clip + slabs - 1) / slabs;
    const int r0 = sl * R, r1 = min(rows_per_clip, r0 + R);
    const float *base = x + (size_t)b * rows_per_clip * C;
    v4f s = {0.f, 0.f, 0.f, 0.f};
    if (rg < rgroups) {
        for (int r = r0 + rg; r < r1; r += rgroups) s += *reinterpret_cast<const v4f *>(base + (size_t)r * C + cg * 4);
        *reinterpret_cast<v4f *>(red + (rg * C4 + cg) * 4) = s;
    }
    __syncthreads();
    if (tid < C4) {
        v4f t = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < rgroups; ++g) t += *reinterpret_cast<const v4f *>(red + (g * C4 + tid) * 4);
        *reinterpret_cast<v4f *>(partial + (size_t)blockIdx.x * C + tid * 4) = t;
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

extern "C" {

int nsg_relu_backward_add(const float *a, const float *b, const float *x, float *dx, int64_t n, void *stream)
{
    NSG_REQUIRE(a && x && dx && n >= 0, NSG_E_INVALID, "nsg_relu_backward_add: bad argument");
    if (n == 0) return NSG_OK;
    hipStream_t s = (hipStream_t)stream;
    if ((n & 3) == 0 && nsg_aligned16(a) && nsg_aligned16(x) && nsg_aligned16(dx) && (!b || nsg_aligned16(b)))
        hipLaunchKernelGGL((relu_bwd_add_kernel<4>), dim3(ew_blocks(n / 4)), dim3(256), 0, s, a, b, x, dx, n / 4);
    else
        hipLaunchKernelGGL((relu_bwd_add_kernel<1>), dim3(ew_blocks(n)), dim3(256), 0, s, a, b, x, dx, n);
    return nsg_check_launch("relu_bwd_add_kernel");
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

int nsg_add_per_clip(const float *x, const float *rows, float *y, int32_t B, int64_t rows_per_clip, int32_t C, void *stream)
{
    NSG_REQUIRE(x && rows && y && B > 0 && rows_per_clip > 0 && C > 0, NSG_E_INVALID, "nsg_add_per_clip: bad argument");
    NSG_REQUIRE(C % 4 == 0 && nsg_aligned16(x) && nsg_aligned16(rows) && nsg_aligned16(y), NSG_E_UNSUPPORTED,
                "nsg_add_per_clip: C must be a multiple of 4 and pointers 16-byte aligned");
    const int64_t n4 = (int64_t)B * rows_per_clip * C / 4;
    hipLaunchKernelGGL(add_per_clip_kernel, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, x, rows, y, n4, C / 4,
                       rows_per_clip * (C / 4));
    return nsg_check_launch("add_per_clip_kernel");
}

size_t nsg_clip_colsum_workspace_bytes(int32_t B, int32_t C) { return (size_t)(B > 0 ? B : 0) * CLIP_SLABS * (C > 0 ? C : 0) * sizeof(float); }

int nsg_clip_colsum(const float *x, int32_t B, int64_t rows_per_clip, int32_t C, float *out, void *workspace,
                    size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && out && B > 0 && rows_per_clip > 0 && C > 0, NSG_E_INVALID, "nsg_clip_colsum: bad argument");
    NSG_REQUIRE(C % 4 == 0 && C <= 1024 && nsg_aligned16(x) && rows_per_clip < (1ll << 31), NSG_E_UNSUPPORTED,
                "nsg_clip_colsum: C must be a multiple of 4 (<= 1024), x 16-byte aligned");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_clip_colsum_workspace_bytes(B, C), NSG_E_WORKSPACE, "nsg_clip_colsum: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float *partial = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(clip_colsum_partial_kernel, dim3(B * CLIP_SLABS), dim3(256), 0, s, x, (int)rows_per_clip, C, CLIP_SLABS, partial);
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

int nsg_vq_losses(const float *z, const float *q, int64_t n, float dz_scale, float dq_scale, const float *dz_add,
                  float *loss_out, float *dz, float *dq, void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(z && q && loss_out && n > 0, NSG_E_INVALID, "nsg_vq_losses: bad argument");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_reduce_workspace_bytes(n), NSG_E_WORKSPACE, "nsg_vq_losses: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    double *partial = reinterpret_cast<double *>(workspace);
    const float zs = dz_scale * 2.0f / (float)n, qs = dq_scale * 2.0f / (float)n;
    const bool vec = (n & 3) == 0 && nsg_aligned16(z) && nsg_aligned16(q) && (!dz || nsg_aligned16(dz)) &&
                     (!dq || nsg_aligned16(dq)) && (!dz_add || nsg_aligned16(dz_add));
    int nb;
    if (vec) {
        nb = ew_blocks(n / 4);
        if (nb > RED_BLOCKS) nb = RED_BLOCKS;
        hipLaunchKernelGGL((vq_losses_kernel<4>), dim3(nb), dim3(256), 0, s, z, q, n / 4, zs, qs, dz_add, dz, dq, partial);
    } else {
        nb = ew_blocks(n);
        if (nb > RED_BLOCKS) nb = RED_BLOCKS;
        hipLaunchKernelGGL((vq_losses_kernel<1>), dim3(nb), dim3(256), 0, s, z, q, n, zs, qs, dz_add, dz, dq, partial);
    }
    hipLaunchKernelGGL(final_mean_kernel, dim3(1), dim3(64), 0, s, partial, nb, (double)n, loss_out);
    return nsg_check_launch("vq_losses");
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
