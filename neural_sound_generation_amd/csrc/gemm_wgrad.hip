// Weight-gradient GEMM (reduction over pixels) on v_mfma_f32_32x32x2_f32, split over pixel slabs,
// with a fixed-order slab reduction (bitwise reproducible; no float atomics).
//
//   G[t][a][c] = sum over pixels m of the conv-OUTPUT grid   P[m][a] * Q[m shifted by tap t][c]
//
//   nn.Conv2d          : P = dy (a = C_out), Q = x  (c = C_in)  ->  dW[a][c][kh][kw]
//   nn.ConvTranspose2d : P = x  (a = C_in),  Q = dy (c = C_out) ->  dW[a][c][kh][kw]
//       (a transposed conv's "conv-output grid" is its own input grid: out pixel = 2*in - 1 + k)
//   one-hot mode       : P[m][a] = (idx[m] == a), single tap, Q = g rows -> index_add_ over codes
//       (autograd of torch.index_select, src/models.py:137 / vector_quantization.py:60-61)
//
// Both operands are consumed "transposed" (the reduction index is the row of the NHWC tensors),
// which is exactly how NHWC rows sit in memory: the LDS tiles are plain [32 pixels][channels]
// copies, and an MFMA operand read is 32 consecutive floats of one pixel row (ds_read_b32,
// conflict-free).  Block = 4 waves, tile TA x TC of one tap for one slab of pixels.
#include "nsg_common.h"

namespace {

constexpr int KP = 32;  // pixels per chunk

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_gemm_f32(const WgradParams p)
{
    constexpr int TA = WM * TM * 32;
    constexpr int TC = WN * TN * 32;
    constexpr int PJ = TA / 32;  // float4 per thread per chunk for P (KP*TA/4/256)
    constexpr int QJ = TC / 32;
    constexpr int PA4 = TA / 4, QC4 = TC / 4;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ps = smem;                 // [2][KP][TA]
    float *Qs = smem + 2 * KP * TA;   // [2][KP][TC]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;

    const int slab = blockIdx.x;
    const int tap = blockIdx.y;
    const int ctiles = (p.C + TC - 1) / TC;
    const int a0 = (blockIdx.z / ctiles) * TA;
    const int c0 = (blockIdx.z % ctiles) * TC;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    const int mbeg = slab * p.slab_rows;
    const int mend = min(p.Mp, mbeg + p.slab_rows);
    const int nchunk = (mend - mbeg + KP - 1) / KP;

    v4f rp[PJ], rq[QJ];

    auto gload = [&](int ch) {
        const int mb = mbeg + ch * KP;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int f = tid + 256 * j;
            const int pix = f / PA4;
            const int a4 = (f - pix * PA4) * 4;
            const int m = mb + pix;
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (m < mend && (a0 + a4) < p.A) {
                if (p.onehot) {
                    const int code = (int)p.idx[m] - (a0 + a4);
                    v.x = code == 0 ? 1.f : 0.f;
                    v.y = code == 1 ? 1.f : 0.f;
                    v.z = code == 2 ? 1.f : 0.f;
                    v.w = code == 3 ? 1.f : 0.f;
                } else {
                    v = *reinterpret_cast<const v4f *>(p.P + ((size_t)m * p.A + a0 + a4));
                    if (p.relu_p) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
            }
            rp[j] = v;
        }
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const int f = tid + 256 * j;
            const int pix = f / QC4;
            const int cc4 = (f - pix * QC4) * 4;
            const int m = mb + pix;
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (m < mend && (c0 + cc4) < p.C) {
                const int b = m / (p.PH * p.PW);
                const int rem = m - b * (p.PH * p.PW);
                const int py = rem / p.PW;
                const int px = rem - py * p.PW;
                const int qy = py * p.stride - p.pad + kh;
                const int qx = px * p.stride - p.pad + kw;
                if (qy >= 0 && qy < p.QH && qx >= 0 && qx < p.QW) {
                    v = *reinterpret_cast<const v4f *>(p.Q + (((size_t)(b * p.QH + qy) * p.QW + qx) * p.C + c0 + cc4));
                    if (p.relu_q) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
            }
            rq[j] = v;
        }
    };
    auto lstore = [&](int buf) {
        float *ps = Ps + buf * KP * TA;
        float *qs = Qs + buf * KP * TC;
#pragma unroll
        for (int j = 0; j < PJ; ++j) *reinterpret_cast<v4f *>(ps + (tid + 256 * j) * 4) = rp[j];
#pragma unroll
        for (int j = 0; j < QJ; ++j) *reinterpret_cast<v4f *>(qs + (tid + 256 * j) * 4) = rq[j];
    };

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nchunk > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nchunk) gload(ch + 1);
        const float *ps = Ps + cur * KP * TA + wr * TM * 32 + l31;
        const float *qs = Qs + cur * KP * TC + wc * TN * 32 + l31;
#pragma unroll
        for (int s = 0; s < KP / 2; ++s) {
            const int k = 2 * s + h;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = ps[k * TA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = qs[k * TC + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (ch + 1 < nchunk) lstore(cur ^ 1);
        __syncthreads();
    }

    const int ntaps = p.KH * p.KW;
    float *dst = p.partial + ((size_t)(slab * ntaps + tap) * p.A) * p.C;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = c0 + wc * TN * 32 + j * 32 + l31;
        if (c >= p.C) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int a = a0 + wr * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (a < p.A) dst[(size_t)a * p.C + c] = acc[i][j][r];
            }
    }
}

// dst[(a*C + c)*ntaps + t] = sum_slab partial[slab][t][a][c]
__global__ void wgrad_reduce_kernel(const float *partial, float *dst, int nslab, int ntaps, int A, int C)
{
    const int64_t total = (int64_t)ntaps * A * C;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int a = (int)((e / C) % A);
        const int t = (int)(e / ((int64_t)A * C));
        float s = 0.f;
        for (int sl = 0; sl < nslab; ++sl) s += partial[(size_t)sl * total + e];
        dst[((size_t)a * C + c) * ntaps + t] = s;
    }
}

struct SlabPlan {
    int nslab;
    int slab_rows;
};

SlabPlan plan_slabs(int64_t Mp, int ntaps, int A, int C)
{
    const int TA = A > 64 ? 128 : (A > 32 ? 64 : 32);  // informational; tiles chosen in dispatch below
    (void)TA;
    const int64_t tiles = nsg_cdiv(A, 128) * nsg_cdiv(C, C > 32 ? 128 : 32) * ntaps;
    int64_t want = nsg_cdiv(1536, tiles);           // aim for ~6 blocks per CU in total
    const int64_t maxslab = nsg_cdiv(Mp, 256);      // at least 256 pixels per slab
    if (want > maxslab) want = maxslab;
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    int64_t rows = nsg_cdiv(nsg_cdiv(Mp, want), KP) * KP;
    SlabPlan sp;
    sp.slab_rows = (int)rows;
    sp.nslab = (int)nsg_cdiv(Mp, rows);
    return sp;
}

template <int WM, int WN, int TM, int TN>
int launch_wg(const WgradParams &p, int nslab, hipStream_t s)
{
    constexpr int TA = WM * TM * 32, TC = WN * TN * 32;
    const size_t lds = (size_t)2 * KP * (TA + TC) * sizeof(float);
    const int ntaps = p.KH * p.KW;
    const int tiles = (int)(nsg_cdiv(p.A, TA) * nsg_cdiv(p.C, TC));
    dim3 grid(nslab, ntaps, tiles);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_gemm_f32<WM, WN, TM, TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return nsg_fail((int)e, "wgrad: cannot reserve %zu bytes of LDS", lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_gemm_f32<WM, WN, TM, TN>), grid, dim3(256), lds, s, p);
    return nsg_check_launch("wgrad_gemm_f32");
}

}  // namespace

size_t nsg_wgrad_workspace_bytes(int64_t Mp, int ntaps, int A, int C)
{
    const SlabPlan sp = plan_slabs(Mp, ntaps, A, C);
    return (size_t)sp.nslab * ntaps * A * C * sizeof(float);
}

int nsg_launch_wgrad(WgradParams p, float *dst, void *ws, size_t ws_bytes, hipStream_t s)
{
    const int ntaps = p.KH * p.KW;
    if (p.Mp <= 0) return nsg_fail(NSG_E_INVALID, "wgrad: empty reduction");
    if (p.A % 4 != 0 || p.C % 4 != 0) return nsg_fail(NSG_E_UNSUPPORTED, "wgrad: channels (%d,%d) must be multiples of 4", p.A, p.C);
    if ((!p.onehot && !nsg_aligned16(p.P)) || !nsg_aligned16(p.Q)) return nsg_fail(NSG_E_INVALID, "wgrad: operands must be 16-byte aligned");
    const SlabPlan sp = plan_slabs(p.Mp, ntaps, p.A, p.C);
    const size_t need = (size_t)sp.nslab * ntaps * p.A * p.C * sizeof(float);
    if (ws == nullptr || ws_bytes < need) return nsg_fail(NSG_E_WORKSPACE, "wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    p.partial = reinterpret_cast<float *>(ws);
    p.slab_rows = sp.slab_rows;
    int rc;
    if (p.C <= 32) {
        rc = launch_wg<4, 1, 1, 1>(p, sp.nslab, s);          // 128 x 32 (im2col'd single-channel layers)
    } else if (p.A <= 64 && p.C <= 64) {
        rc = launch_wg<2, 2, 1, 1>(p, sp.nslab, s);          // 64 x 64
    } else {
        rc = launch_wg<2, 2, 2, 2>(p, sp.nslab, s);          // 128 x 128
    }
    if (rc != NSG_OK) return rc;
    const int64_t total = (int64_t)ntaps * p.A * p.C;
    const int blocks = (int)(nsg_cdiv(total, 256) > 2048 ? 2048 : nsg_cdiv(total, 256));
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.partial, dst, sp.nslab, ntaps, p.A, p.C);
    return nsg_check_launch("wgrad_reduce_kernel");
}
