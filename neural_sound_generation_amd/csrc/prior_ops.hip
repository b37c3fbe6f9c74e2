// Element-wise pieces of the latent prior (the reference's GatedPixelCNN over the code-index grid, src/models.py:219-341,
// SURVEY.md section 8f row 1); its convolutions run on the conv kernels of the main path.
//   gated activation  y = tanh(a) * sigmoid(b),  (a, b) = the two channel halves of x (+ a per-clip conditioning row)
//                     src/models.py:219-226 (GatedActivation) with the class embedding add of :268,:274 folded in;
//   cross-entropy     mean over rows of  logsumexp(l) - l[target]  and its gradient (what F.cross_entropy computes on
//                     the prior's logits).
// fp32, NHWC rows [M][channels]; deterministic (fixed-order reductions).
#include "nsg_common.h"
#include <math.h>

namespace {

// x [M][2C], cond [B][2C] or null (row m belongs to clip m / rows_per_clip), y [M][C]
__global__ __launch_bounds__(256) void gated_fwd_kernel(const float *__restrict__ x, const float *__restrict__ cond, float *__restrict__ y,
                                                        int64_t M, int C, int64_t rows_per_clip)
{
    const int C4 = C >> 2;
    const int64_t total = M * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        v4f a = *reinterpret_cast<const v4f *>(x + m * 2 * C + c);
        v4f b = *reinterpret_cast<const v4f *>(x + m * 2 * C + C + c);
        if (cond) {
            const float *cr = cond + (m / rows_per_clip) * 2 * C;
            a += *reinterpret_cast<const v4f *>(cr + c);
            b += *reinterpret_cast<const v4f *>(cr + C + c);
        }
        v4f o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = tanhf(a[e]) * (1.f / (1.f + expf(-b[e])));
        *reinterpret_cast<v4f *>(y + m * C + c) = o;
    }
}

// dx [M][2C] from dy [M][C]:  d/da = dy * s * (1 - t^2),  d/db = dy * t * s * (1 - s)
__global__ __launch_bounds__(256) void gated_bwd_kernel(const float *__restrict__ x, const float *__restrict__ cond,
                                                        const float *__restrict__ dy, float *__restrict__ dx, int64_t M, int C,
                                                        int64_t rows_per_clip)
{
    const int C4 = C >> 2;
    const int64_t total = M * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        v4f a = *reinterpret_cast<const v4f *>(x + m * 2 * C + c);
        v4f b = *reinterpret_cast<const v4f *>(x + m * 2 * C + C + c);
        if (cond) {
            const float *cr = cond + (m / rows_per_clip) * 2 * C;
            a += *reinterpret_cast<const v4f *>(cr + c);
            b += *reinterpret_cast<const v4f *>(cr + C + c);
        }
        const v4f g = *reinterpret_cast<const v4f *>(dy + m * C + c);
        v4f da, db;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = tanhf(a[e]);
            const float s = 1.f / (1.f + expf(-b[e]));
            da[e] = g[e] * s * (1.f - t * t);
            db[e] = g[e] * t * s * (1.f - s);
        }
        *reinterpret_cast<v4f *>(dx + m * 2 * C + c) = da;
        *reinterpret_cast<v4f *>(dx + m * 2 * C + C + c) = db;
    }
}

// One wave per row of K logits.  row_loss[m] = logsumexp - l[target];  dlogits = (softmax - onehot) * gscale (optional).
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float *__restrict__ logits, const int64_t *__restrict__ target,
                                                            int64_t M, int K, float gscale, float *__restrict__ row_loss,
                                                            float *__restrict__ dlogits)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t m = wave; m < M; m += nw) {
        const float *l = logits + m * K;
        float mx = -INFINITY;
        for (int k = lane; k < K; k += 64) mx = fmaxf(mx, l[k]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float se = 0.f;
        for (int k = lane; k < K; k += 64) se += expf(l[k] - mx);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) se += __shfl_xor(se, off, 64);   // xor butterfly: the same value, in the same order, on every lane
        const float lse = mx + logf(se);
        const int t = (int)target[m];
        if (lane == 0) row_loss[m] = (t >= 0 && t < K) ? lse - l[t] : 0.f;
        if (dlogits) {
            const float inv = 1.f / se;
            for (int k = lane; k < K; k += 64) dlogits[m * K + k] = (expf(l[k] - mx) * inv - (k == t ? 1.f : 0.f)) * gscale;
        }
    }
}

// sum of n floats in double, two fixed-order stages
__global__ __launch_bounds__(256) void sum_partial_kernel(const float *__restrict__ v, int64_t n, double *__restrict__ partial)
{
    __shared__ double red[256];
    double acc = 0.0;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = blockIdx.x * per, i1 = min(n, i0 + per);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) acc += (double)v[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 256; ++k) t += red[k];
        partial[blockIdx.x] = t;
    }
}
__global__ void sum_final_kernel(const double *partial, int n, double denom, float *out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n; ++i) t += partial[i];
        out[0] = (float)(t / denom);
    }
}

constexpr int SUM_BLOCKS = 256;
inline int ew_blocks(int64_t n) { const int64_t b = nsg_cdiv(n, 256); return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" {

int nsg_gated_activation_forward(const float *x, const float *cond, float *y, int64_t M, int32_t C, int64_t rows_per_clip, void *stream)
{
    NSG_REQUIRE(x && y && M > 0 && C > 0 && C % 4 == 0, NSG_E_INVALID, "nsg_gated_activation_forward: bad argument (C %% 4 == 0)");
    NSG_REQUIRE(!cond || rows_per_clip > 0, NSG_E_INVALID, "nsg_gated_activation_forward: rows_per_clip must be positive with a conditioning row");
    NSG_REQUIRE(nsg_aligned16(x) && nsg_aligned16(y) && (!cond || nsg_aligned16(cond)), NSG_E_INVALID, "nsg_gated_activation_forward: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(gated_fwd_kernel, dim3(ew_blocks(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, cond, y, M, C, cond ? rows_per_clip : 1);
    return nsg_check_launch("gated_fwd_kernel");
}

int nsg_gated_activation_backward(const float *x, const float *cond, const float *dy, float *dx, int64_t M, int32_t C, int64_t rows_per_clip,
                                  void *stream)
{
    NSG_REQUIRE(x && dy && dx && M > 0 && C > 0 && C % 4 == 0, NSG_E_INVALID, "nsg_gated_activation_backward: bad argument (C %% 4 == 0)");
    NSG_REQUIRE(!cond || rows_per_clip > 0, NSG_E_INVALID, "nsg_gated_activation_backward: rows_per_clip must be positive with a conditioning row");
    NSG_REQUIRE(nsg_aligned16(x) && nsg_aligned16(dy) && nsg_aligned16(dx) && (!cond || nsg_aligned16(cond)), NSG_E_INVALID,
                "nsg_gated_activation_backward: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(gated_bwd_kernel, dim3(ew_blocks(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, cond, dy, dx, M, C, cond ? rows_per_clip : 1);
    return nsg_check_launch("gated_bwd_kernel");
}

size_t nsg_cross_entropy_workspace_bytes(int64_t M) { return M > 0 ? nsg_align_up((size_t)M * sizeof(float), 256) + SUM_BLOCKS * sizeof(double) : 0; }

int nsg_cross_entropy(const float *logits, const int64_t *target, int64_t M, int32_t K, float grad_scale, float *loss_out, float *dlogits,
                      void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(logits && target && loss_out && M > 0 && K > 0, NSG_E_INVALID, "nsg_cross_entropy: bad argument");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_cross_entropy_workspace_bytes(M), NSG_E_WORKSPACE, "nsg_cross_entropy: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float *row_loss = reinterpret_cast<float *>(workspace);
    double *partial = reinterpret_cast<double *>(reinterpret_cast<char *>(workspace) + nsg_align_up((size_t)M * sizeof(float), 256));
    int64_t blocks = nsg_cdiv(M, 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cross_entropy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, logits, target, M, K, grad_scale / (float)M, row_loss, dlogits);
    const int nb = (int)(M < SUM_BLOCKS ? M : SUM_BLOCKS);
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nb), dim3(256), 0, s, row_loss, M, partial);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(64), 0, s, partial, nb, (double)M, loss_out);
    return nsg_check_launch("cross_entropy");
}

}  // extern "C"
