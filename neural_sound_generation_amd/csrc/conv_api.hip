// Convolution entry points: geometry checks, weight re-packing, the single-channel (C == 1)
// col2im staging, bias gradients, and dispatch onto the two GEMM kernels.
// Reference call sites: src/models.py:150,153,165,168 (nn.Conv2d), :179,182 (nn.ConvTranspose2d).
#include "nsg_common.h"

// diagnostics: when set, the next gather_gemm launches stamp their main-loop clocks into this buffer
NSG_DIAG_SWITCH(unsigned long long *, g_debug_stamps, nullptr)     // diagnostics library only (-DNSG_DIAG)
#ifdef NSG_DIAG
extern "C" NSG_API void nsg_debug_set_stamp_buffer(unsigned long long *buf) { g_debug_stamps = buf; }
#endif

// stencil_c1.hip
bool nsg_c1_stencil_supported(int C);
size_t nsg_c1_stencil_wgrad_workspace_bytes(int C);
int nsg_launch_c1_stencil_fwd(const float *img, const float *w, const float *bias, void *out, int out_dtype, int B, int LH, int LW,
                              int HH, int WW, int C, hipStream_t s);
// gemm_flat.hip: the ResBlock's 1x1 conv as a flat GEMM with BatchNorm work in its operand staging (nsg_bn_relu_conv1x1_*)
bool nsg_flat1x1_supported(int dtype, int C);
size_t nsg_flat1x1_workspace_bytes(int C);
int nsg_launch_flat1x1_forward(const void *x, const float *mean, const float *invstd, const float *gamma, const float *beta, const float *w,
                               const float *bias, void *y, int64_t M, int C, void *ws, int want_stats, int *nblocks, hipStream_t s);
int nsg_launch_flat1x1_backward(const void *h, const void *dy, const float *mean, const float *invstd, const float *gamma, const float *dgamma,
                                const float *dbeta, const float *w, void *dh, void *dx, int64_t M, int C, void *ws, int *nblocks,
                                const void *prev_x, const float *prev_mean, const float *prev_invstd, const float *prev_gamma,
                                const float *prev_beta, float **prev_partial, hipStream_t s);
constexpr int C1_LOSS_BLOCKS = 1024;
int nsg_launch_final_mean(const double *partial, int n, double denom, float *out, hipStream_t s);     // elementwise.hip
bool nsg_flat1x1_fused_bwd_supported(int dtype, int C);
size_t nsg_flat1x1_fused_bwd_workspace_bytes(int C);
int nsg_launch_flat1x1_fused_bwd(const void *h, const void *dy, const float *mean, const float *invstd, const float *gamma, const float *dgamma,
                                 const float *dbeta, const float *w, void *dx, float *dw, int64_t M, int C, void *ws, int *nblocks,
                                 const void *prev_x, const float *prev_mean, const float *prev_invstd, const float *prev_gamma,
                                 const float *prev_beta, float **colsum_partial, float **prev_partial, float **dw_partial, hipStream_t s);
// c1_mfma.hip / stencil_c1.hip: pieces of the fused output layer (nsg_bn_relu_c1convt_*)
bool nsg_c1m_supported(int C);
int nsg_launch_bnrelu_dots(const void *u, const float *mean, const float *invstd, const float *gamma, const float *beta, const float *w,
                           float *dots, int64_t M, int C, hipStream_t s);
int nsg_launch_c1m_out_bwd_sums(const float *dimg, const float *w, const void *u, const float *mean, const float *invstd, const float *gamma,
                                const float *beta, float *partial, int blocks, int B, int LH, int LW, int C, hipStream_t s);
int nsg_launch_c1m_out_bwd_apply(const float *dimg, const float *w, const void *u, const float *mean, const float *invstd, const float *gamma,
                                 const float *beta, const float *dgamma, const float *dbeta, float inv_m, float *partial17, void *du, int blocks,
                                 int B, int LH, int LW, int C, hipStream_t s);
int nsg_launch_c1_stencil_wgrad_final(const float *partial17, int blocks, int C, float *dw, float *colsum, hipStream_t s);
int nsg_launch_c1_stencil_wgrad(const float *img, const void *t, int t_dtype, int relu_t, float *dw, float *colsum, int B, int LH,
                                int LW, int HH, int WW, int C, void *ws, size_t ws_bytes, hipStream_t s);

namespace {

// One weight re-pack:  dst[(t*NN + n)*CC + c] = src[n*sn + c*sc + (flip ? T-1-t : t)]  (dst fp32 or bf16)
// frag (bf16 images with NN % 128 == 0 and CC % 64 == 0 only): a SECOND image behind the first (T*NN*CC elements further) in
// the order gemm_patch.hip's waves load their MFMA operand fragments -- [t][c / 64][n / 32][(c % 64) / 16][lane][8] with
// lane = n % 32 + 32 * ((c % 16) / 8): one wave-wide 16-byte load = 1 KiB contiguous (nsg_frag_image_index).
struct PackJob {
    const float *src;
    void *dst;
    int T, NN, CC, sn, sc, flip, bf16, frag;
};
constexpr int PACK_MAX_JOBS = 32;
struct PackJobs {
    PackJob job[PACK_MAX_JOBS];
};
// blockIdx.y = job: every packed image of a training step in ONE launch
__global__ __launch_bounds__(256) void pack_w_kernel(const PackJobs jobs)
{
    const PackJob j = jobs.job[blockIdx.y];
    const int64_t total = (int64_t)j.T * j.NN * j.CC;
    if (j.bf16 && j.CC % 8 == 0 && (reinterpret_cast<uintptr_t>(j.dst) & 15u) == 0) {
        // 8 consecutive c per thread: 8 independent (strided) reads in flight, one 16-byte store per image (8 consecutive c are
        // contiguous in the fragment order too) instead of eight 2-byte ones
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const int CG = j.CC / 8;
        for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < total / 8; g += (int64_t)gridDim.x * blockDim.x) {
            const int c = (int)(g % CG) * 8;
            const int n = (int)((g / CG) % j.NN);
            const int t = (int)(g / ((int64_t)CG * j.NN));
            const float *sp = j.src + (size_t)n * j.sn + (size_t)c * j.sc + (j.flip ? j.T - 1 - t : t);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = sp[(size_t)e * j.sc];
            u4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = nsg_pack_bf16(v[2 * e], v[2 * e + 1]);
            bf16_t *d = reinterpret_cast<bf16_t *>(j.dst);
            *reinterpret_cast<u4 *>(d + ((size_t)(t * j.NN + n) * j.CC + c)) = pk;
            if (j.frag) *reinterpret_cast<u4 *>(d + total + nsg_frag_index(t, n, c, j.NN, j.CC)) = pk;
        }
        return;
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % j.CC);
        const int n = (int)((i / j.CC) % j.NN);
        const int t = (int)(i / ((int64_t)j.CC * j.NN));
        const float v = j.src[(size_t)n * j.sn + (size_t)c * j.sc + (j.flip ? j.T - 1 - t : t)];
        if (j.bf16) {
            reinterpret_cast<bf16_t *>(j.dst)[i] = nsg_f2bf(v);
            if (j.frag) {
                reinterpret_cast<bf16_t *>(j.dst)[total + nsg_frag_index(t, n, c, j.NN, j.CC)] = nsg_f2bf(v);
            }
        } else {
            reinterpret_cast<float *>(j.dst)[i] = v;
        }
    }
}

// blocks of col2im_c1_kernel for a (B, LH, LW) record grid: one per tile, at most C1_LOSS_BLOCKS (they walk the tiles)
constexpr int C1_COL2IM_TY = 4, C1_COL2IM_TX = 64;      // records per tile = threads per block; the model's grids (20 x 256, 40 x 512) tile exactly
inline int col2im_blocks(int B, int HH, int WW)        // (HH, WW): the image; positions = 2 x 2 output blocks
{
    const int64_t nt = (int64_t)B * (((HH + 1) / 2 + C1_COL2IM_TY - 1) / C1_COL2IM_TY) * (((WW + 1) / 2 + C1_COL2IM_TX - 1) / C1_COL2IM_TX);
    return (int)(nt < C1_LOSS_BLOCKS ? (nt < 1 ? 1 : nt) : C1_LOSS_BLOCKS);
}

// dots [B][LH][LW][16] -> out [B][HH][WW]:  out[y][x] = bias + sum of the (<= 4) taps that reach it (y = 2 ly - 1 + kh, x = 2 lx - 1 + kw).
// A thread owns the ALIGNED 2 x 2 outputs (2 ly .. 2 ly + 1) x (2 lx .. 2 lx + 1) of position (ly, lx), ly < ceil(HH / 2), lx <
// ceil(WW / 2) (one past the record grid when the image extent is odd: its record is a zero): 8-byte loads of the
// target and 8-byte stores, a wave = 512 contiguous bytes of an image row (owning the 2 x 2 between four records instead, every
// access was a 4-byte one at a stride of 8).  Each of the four takes one tap from four of the 3 x 3 records around (ly, lx):
//   (2ly,   2lx)   [1][1] of (ly, lx), [1][3] of (ly, lx-1), [3][1] of (ly-1, lx), [3][3] of (ly-1, lx-1)
//   (2ly,   2lx+1) [1][2] of (ly, lx), [1][0] of (ly, lx+1), [3][2] of (ly-1, lx), [3][0] of (ly-1, lx+1)
//   (2ly+1, 2lx)   [2][1] of (ly, lx), [2][3] of (ly, lx-1), [0][1] of (ly+1, lx), [0][3] of (ly+1, lx-1)
//   (2ly+1, 2lx+1) [2][2] of (ly, lx), [2][0] of (ly, lx+1), [0][2] of (ly+1, lx), [0][0] of (ly+1, lx+1)
// added from the record with the largest (row, column) down, as before.  A block walks tiles of C2I_TY x C2I_TX positions; the
// (TY + 2) x (TX + 2) records a tile touches come in whole through LDS (16-byte pieces, consecutive threads on consecutive
// pieces; records outside the grid are zeros), the next tile's while this one is worked on (a tile is one global-memory latency
// otherwise).  Measured in the step at 128 clips (252 MB): 73.7 us gathered from global memory -> 62.6 staged + prefetched ->
// 55-57 with the aligned 2 x 2 blocks.
// LOSS: the reconstruction loss and its gradient in the same pass (train.py:118-129 on x_tilde = tanh(out)): per block a partial
// of sum (pad(x_tilde) - target)^2 over the target's T >= WW columns, dpre = gscale (x_tilde - target)(1 - x_tilde^2) = the
// gradient w.r.t. the tanh INPUT; x_tilde itself is stored only when out != null.
// tanh by one exponential: (1 - t) / (1 + t), t = exp(-2 |x|) in (0, 1]: no overflow, absolute error ~1e-7 (the image is compared
// and consumed by absolute value)
__device__ __forceinline__ float c1_tanh(float x)
{
    const float t = __expf(-2.f * fabsf(x));
    return copysignf(__fdividef(1.f - t, 1.f + t), x);
}
constexpr int C2I_TY = C1_COL2IM_TY, C2I_TX = C1_COL2IM_TX;
static_assert(C2I_TY * C2I_TX == 256, "one thread per position");
template <bool LOSS>
__global__ __launch_bounds__(256) void col2im_c1_kernel(const float *__restrict__ dots, const float *__restrict__ bias, float *__restrict__ out,
                                                        int B, int LH, int LW, int HH, int WW, int tanh_out, const float *__restrict__ target,
                                                        int T, float gscale, float *__restrict__ dpre, double *__restrict__ partial)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int RY = C2I_TY + 2, RX = C2I_TX + 2, RP = 16 + 4;      // records per tile; floats per record in LDS (80 bytes: b128 writes,
    __shared__ __attribute__((aligned(16))) float rec[RY * RX * RP];  //  consecutive records 20 banks apart)
    const int PY = (HH + 1) / 2, PX = (WW + 1) / 2;      // positions
    const int tiles_y = (PY + C2I_TY - 1) / C2I_TY, tiles_x = (PX + C2I_TX - 1) / C2I_TX;
    const int64_t ntiles = (int64_t)B * tiles_y * tiles_x;
    const int tid = threadIdx.x;
    const int py = tid / C2I_TX, px = tid - py * C2I_TX;
    const float bv = bias ? bias[0] : 0.f;
    const bool vec = (T & 1) == 0, vec_out = (WW & 1) == 0;      // 8-byte accesses need an even row pitch
    double lacc = 0.0, gacc = 0.0;      // loss partial; sum of dpre (= the transposed conv's bias gradient)
    constexpr int NPC = (RY * RX * 4 + 255) / 256;      // 16-byte pieces of a tile's records per thread
    v4f pc[NPC];
    auto tile_coords = [&](int64_t tile, int &b, int &ly0, int &lx0) {
        const int tx = (int)(tile % tiles_x);
        const int ty = (int)((tile / tiles_x) % tiles_y);
        b = (int)(tile / ((int64_t)tiles_x * tiles_y));
        ly0 = ty * C2I_TY;
        lx0 = tx * C2I_TX;
    };
    auto fetch = [&](int64_t tile) {        // the tile's records and their halo -> registers (zeros outside the grid)
        int b, ly0, lx0;
        tile_coords(tile, b, ly0, lx0);
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int q = tid + 256 * i;
            const int r = min(q >> 2, RY * RX - 1), piece = q & 3;
            const int ry = r / RX, rx = r - ry * RX;
            const int y = ly0 - 1 + ry, x = lx0 - 1 + rx;
            const bool ok = y >= 0 && y < LH && x >= 0 && x < LW;
            const v4f v = *reinterpret_cast<const v4f *>(dots + (((size_t)b * LH + (ok ? y : 0)) * LW + (ok ? x : 0)) * 16 + 4 * piece);   // clamped, unconditional
            pc[i] = ok ? v : v4f{0.f, 0.f, 0.f, 0.f};
        }
    };
    if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b, ly0, lx0;
        tile_coords(tile, b, ly0, lx0);
        __syncthreads();                // the previous tile's records have been read
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int q = tid + 256 * i;
            if (q < RY * RX * 4) *reinterpret_cast<v4f *>(rec + (q >> 2) * RP + 4 * (q & 3)) = pc[i];
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        const int ly = ly0 + py, lx = lx0 + px;
        if (ly >= PY || lx >= PX) continue;     // (positions past the image; every thread reaches the next round's barriers)
        // the 3 x 3 records around (ly, lx): row m, column n <-> record (ly - 1 + m, lx - 1 + n); tap [kh][kw] = float 4 kh + kw
        const float *rc = rec + ((py + 1) * RX + (px + 1)) * RP;
        auto tap = [&](int m, int n, int kh, int kw) { return rc[((m - 1) * RX + (n - 1)) * RP + 4 * kh + kw]; };
        float o[2][2];      // [row 2 ly + j][column 2 lx + k]   (records outside the grid are zeros: they add nothing)
        o[0][0] = (((bv + tap(1, 1, 1, 1)) + tap(1, 0, 1, 3)) + tap(0, 1, 3, 1)) + tap(0, 0, 3, 3);
        o[0][1] = (((bv + tap(1, 2, 1, 0)) + tap(1, 1, 1, 2)) + tap(0, 2, 3, 0)) + tap(0, 1, 3, 2);
        o[1][0] = (((bv + tap(2, 1, 0, 1)) + tap(2, 0, 0, 3)) + tap(1, 1, 2, 1)) + tap(1, 0, 2, 3);
        o[1][1] = (((bv + tap(2, 2, 0, 0)) + tap(2, 1, 0, 2)) + tap(1, 2, 2, 0)) + tap(1, 1, 2, 2);
        float l4 = 0.f, g4 = 0.f;       // this position's four terms in fp32, then one double add each
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int yo = 2 * ly + j, xo = 2 * lx;
            if (yo >= HH) continue;
            const bool two = xo + 1 < WW;       // (an odd image width: the last position holds one column)
            const size_t io = ((size_t)b * HH + yo) * WW + xo;
            const float v0 = tanh_out ? c1_tanh(o[j][0]) : o[j][0], v1 = tanh_out ? c1_tanh(o[j][1]) : o[j][1];
            if (out) {
                if (vec_out) *reinterpret_cast<f2 *>(out + io) = f2{v0, v1};
                else { out[io] = v0; if (two) out[io + 1] = v1; }
            }
            if (LOSS) {
                const float *tp = target + ((size_t)b * HH + yo) * T + xo;
                f2 tv;
                if (vec) tv = *reinterpret_cast<const f2 *>(tp);        // (T >= WW, both even: inside the row)
                else { tv.x = tp[0]; tv.y = two ? tp[1] : 0.f; }
                const float d0 = v0 - tv.x, d1 = two ? v1 - tv.y : 0.f;
                l4 = __builtin_fmaf(d1, d1, __builtin_fmaf(d0, d0, l4));
                const float gp0 = gscale * d0 * (1.f - v0 * v0), gp1 = two ? gscale * d1 * (1.f - v1 * v1) : 0.f;
                if (vec_out) *reinterpret_cast<f2 *>(dpre + io) = f2{gp0, gp1};
                else { dpre[io] = gp0; if (two) dpre[io + 1] = gp1; }
                g4 += gp0;
                g4 += gp1;
                if (lx == PX - 1)              // the target's columns past the image: (0 - c)^2, no gradient
                    for (int xx = WW; xx < T; ++xx) {
                        const float cv = target[((size_t)b * HH + yo) * T + xx];
                        lacc += (double)(cv * cv);
                    }
            }
        }
        lacc += (double)l4;
        gacc += (double)g4;
    }
    if (LOSS) {       // block-level sum in double, fixed order (elementwise.hip: block_sum_store)
        __shared__ double red[256];
        red[threadIdx.x] = lacc;
        __syncthreads();
        if (threadIdx.x < 64) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int q = 0; q < 64; ++q) t += red[q];
            partial[blockIdx.x] = t;
        }
        __syncthreads();
        red[threadIdx.x] = gacc;
        __syncthreads();
        if (threadIdx.x < 64) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int q = 0; q < 64; ++q) t += red[q];
            partial[gridDim.x + blockIdx.x] = t;
        }
    }
}

// column sums of [M][C] in two deterministic stages
constexpr int CS_MAX_SLABS = 512;
template <typename TI>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const TI *__restrict__ x, int64_t M, int C, int slab_rows,
                                                             float *__restrict__ partial)
{
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int Cb = C < 256 ? C : 256;
    const int rgroups = 256 / Cb;
    const int cl = tid % Cb, rg = tid / Cb;
    const int64_t r0 = (int64_t)blockIdx.x * slab_rows;
    const int64_t r1 = min(M, r0 + slab_rows);
    for (int cb = 0; cb < C; cb += Cb) {
        const int c = cb + cl;
        float s = 0.f;
        if (rg < rgroups && c < C)
            for (int64_t r = r0 + rg; r < r1; r += rgroups) s += Elem<TI>::get(x + r * C + c);
        red[tid] = (rg < rgroups) ? s : 0.f;
        __syncthreads();
        if (tid < Cb && cb + tid < C) {
            float t = 0.f;
            for (int g = 0; g < rgroups; ++g) t += red[g * Cb + tid];
            partial[(size_t)blockIdx.x * C + cb + tid] = t;
        }
        __syncthreads();
    }
}
// vector form of the above for C % W == 0 (W = elements per 16 bytes): thread (cg, rg) owns W channels of every rgroups-th row
template <typename TI>
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const TI *__restrict__ x, int64_t M, int C, int slab_rows,
                                                                 float *__restrict__ partial)
{
    constexpr int W = Elem<TI>::N;
    __shared__ __attribute__((aligned(16))) float red[256 * W];
    const int CW = C / W;
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * slab_rows;
    const int64_t r1 = min(M, r0 + slab_rows);
    for (int cb = 0; cb < CW; cb += 256) {          // C <= 256*W -> a single pass
        const int cw = min(256, CW - cb);
        const int rgroups = 256 / cw;
        const int cg = tid % cw, rg = tid / cw;
        float s[W];
#pragma unroll
        for (int e = 0; e < W; ++e) s[e] = 0.f;
        if (rg < rgroups) {
#pragma unroll 4
            for (int64_t r = r0 + rg; r < r1; r += rgroups) {
                float v[W];
                Elem<TI>::load16(x + r * C + (cb + cg) * W, v);
#pragma unroll
                for (int e = 0; e < W; ++e) s[e] += v[e];
            }
#pragma unroll
            for (int e = 0; e < W; ++e) red[(rg * cw + cg) * W + e] = s[e];
        }
        __syncthreads();
        if (tid < cw) {
#pragma unroll
            for (int e = 0; e < W; ++e) {
                float t = 0.f;
                for (int g = 0; g < rgroups; ++g) t += red[(g * cw + tid) * W + e];
                partial[(size_t)blockIdx.x * C + (cb + tid) * W + e] = t;
            }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ partial, int nslab, int C, float *out)
{
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int j = tid & 31;
    const int c = blockIdx.x * 8 + (tid >> 5);
    double s = 0.0;
    if (c < C) {
        const int per = (nslab + 31) / 32;
        const int b0 = j * per, b1 = min(nslab, b0 + per);
        if (b1 > b0) s = nsg_strided_sum<double>(partial + (size_t)b0 * C + c, (size_t)C, b1 - b0);
    }
    red[tid] = s;
    __syncthreads();
    if (j != 0 || c >= C) return;
    s = 0.0;
    for (int k = 0; k < 32; ++k) s += red[tid + k];
    out[c] = (float)s;
}

struct CsGeom { int nslab, rows; };
inline CsGeom cs_geom(int64_t M)
{
    CsGeom g;
    int64_t n = nsg_cdiv(M, 512);
    if (n > CS_MAX_SLABS) n = CS_MAX_SLABS;
    if (n < 1) n = 1;
    g.rows = (int)nsg_cdiv(M, n);
    g.nslab = (int)nsg_cdiv(M, g.rows);
    return g;
}
inline size_t colsum_ws_bytes(int64_t M, int C) { return nsg_align_up((size_t)cs_geom(M).nslab * C * sizeof(float), 256); }

int colsum(const void *x, int dtype, int64_t M, int C, float *out, void *ws, hipStream_t s)
{
    const CsGeom g = cs_geom(M);
    float *partial = reinterpret_cast<float *>(ws);
    if (dtype == NSG_BF16 && C % 8 == 0 && nsg_aligned16(x))
        hipLaunchKernelGGL((colsum_partial_vec_kernel<bf16_t>), dim3(g.nslab), dim3(256), 0, s, reinterpret_cast<const bf16_t *>(x), M, C, g.rows, partial);
    else if (dtype == NSG_BF16)
        hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), dim3(g.nslab), dim3(256), 0, s, reinterpret_cast<const bf16_t *>(x), M, C, g.rows, partial);
    else if (C % 4 == 0 && nsg_aligned16(x))
        hipLaunchKernelGGL((colsum_partial_vec_kernel<float>), dim3(g.nslab), dim3(256), 0, s, reinterpret_cast<const float *>(x), M, C, g.rows, partial);
    else
        hipLaunchKernelGGL((colsum_partial_kernel<float>), dim3(g.nslab), dim3(256), 0, s, reinterpret_cast<const float *>(x), M, C, g.rows, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 7) / 8), dim3(256), 0, s, partial, g.nslab, C, out);
    return nsg_check_launch("colsum");
}

inline int ew_blocks(int64_t n)
{
    int64_t b = nsg_cdiv(n, 256);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

enum Kind { K_CONV, K_CONVT, K_CONV_C1, K_CONVT_C1 };

// kernel extent / padding per dimension: k, pad are the height's (and the width's for a square kernel, k_w == 0)
inline int kh_of(const nsg_conv_desc *d) { return d->k; }
inline int kw_of(const nsg_conv_desc *d) { return d->k_w > 0 ? d->k_w : d->k; }
inline int ph_of(const nsg_conv_desc *d) { return d->pad; }
inline int pw_of(const nsg_conv_desc *d) { return d->k_w > 0 ? d->pad_w : d->pad; }
inline bool rect(const nsg_conv_desc *d) { return d->k_w > 0 && (d->k_w != d->k || d->pad_w != d->pad); }

// validates the descriptor; returns <0 on error, else the layer kind
int classify(const nsg_conv_desc *d, const char *fn)
{
    if (!d) return nsg_fail(NSG_E_INVALID, "%s: null descriptor", fn);
    if (d->B <= 0 || d->IH <= 0 || d->IW <= 0 || d->OH <= 0 || d->OW <= 0 || d->C_in <= 0 || d->C_out <= 0 || d->k <= 0)
        return nsg_fail(NSG_E_INVALID, "%s: non-positive geometry", fn);
    if (d->dtype != NSG_F32 && d->dtype != NSG_BF16) return nsg_fail(NSG_E_INVALID, "%s: unknown dtype %d", fn, d->dtype);
    const int cm = d->dtype == NSG_BF16 ? 8 : 4;   // channel granularity of the 16-byte accesses
    const int64_t nin = (int64_t)d->B * d->IH * d->IW * d->C_in, nout = (int64_t)d->B * d->OH * d->OW * d->C_out;
    if (nin >= (1ll << 31) || nout >= (1ll << 31)) return nsg_fail(NSG_E_UNSUPPORTED, "%s: tensor with >= 2^31 elements", fn);
    if (!d->transposed) {
        const int kh = kh_of(d), kw = kw_of(d), ph = ph_of(d), pw = pw_of(d);
        if (kh > 7 || kw > 7 || (d->stride != 1 && d->stride != 2) || ph < 0 || ph >= kh || pw < 0 || pw >= kw || d->k_w < 0)
            return nsg_fail(NSG_E_UNSUPPORTED, "%s: Conv2d kernel %dx%d stride=%d pad=(%d,%d) not supported", fn, kh, kw, d->stride, ph, pw);
        const int full_h = (d->IH + 2 * ph - kh) / d->stride + 1, full_w = (d->IW + 2 * pw - kw) / d->stride + 1;
        if (d->stride == 1) {   // a stride-1 output may be cropped at the bottom / right (the masked stacks of the prior, models.py:268-273)
            if (d->OH > full_h || d->OW > full_w) return nsg_fail(NSG_E_INVALID, "%s: output extent exceeds the Conv2d geometry", fn);
        } else if (d->OH != full_h || d->OW != full_w || rect(d)) {
            return nsg_fail(NSG_E_INVALID, "%s: output extent does not match Conv2d geometry", fn);
        }
        if (d->C_in == 1) {
            if (rect(d)) return nsg_fail(NSG_E_UNSUPPORTED, "%s: rectangular kernels need C_in > 1", fn);
            if (!(d->k == 4 && d->stride == 2 && d->pad == 1 && d->C_out % cm == 0 && nsg_c1_stencil_supported(d->C_out)))
                return nsg_fail(NSG_E_UNSUPPORTED, "%s: C_in=1 needs k=4,stride=2,pad=1, C_out%%%d==0 and C_out<=1024", fn, cm);
            return K_CONV_C1;
        }
        if (d->C_in % cm || d->C_out % cm) return nsg_fail(NSG_E_UNSUPPORTED, "%s: channels must be multiples of %d", fn, cm);
        if (d->stride == 2 && !(d->k == 4 && d->pad == 1))
            return nsg_fail(NSG_E_UNSUPPORTED, "%s: stride-2 Conv2d needs k=4,pad=1", fn);
        return K_CONV;
    }
    if (!(d->k == 4 && d->stride == 2 && d->pad == 1) || rect(d)) return nsg_fail(NSG_E_UNSUPPORTED, "%s: ConvTranspose2d needs k=4,stride=2,pad=1", fn);
    if (d->OH != 2 * d->IH || d->OW != 2 * d->IW) return nsg_fail(NSG_E_INVALID, "%s: output extent does not match ConvTranspose2d geometry", fn);
    if (d->C_out == 1) {
        if (d->C_in % cm || !nsg_c1_stencil_supported(d->C_in))
            return nsg_fail(NSG_E_UNSUPPORTED, "%s: C_in must be a multiple of %d and <= 1024", fn, cm);
        return K_CONVT_C1;
    }
    if (d->C_in % cm || d->C_out % cm) return nsg_fail(NSG_E_UNSUPPORTED, "%s: channels must be multiples of %d", fn, cm);
    return K_CONVT;
}

inline size_t esize(const nsg_conv_desc *d) { return d->dtype == NSG_BF16 ? 2 : 4; }
inline int64_t lowres_pixels(const nsg_conv_desc *d)
{
    return d->transposed ? (int64_t)d->B * d->IH * d->IW : (int64_t)d->B * d->OH * d->OW;
}
// staging image of the single-channel layers: [low-res pixels][16]: patches (dtype) or tap products (fp32)
inline size_t patches_bytes(const nsg_conv_desc *d) { return nsg_align_up((size_t)lowres_pixels(d) * 16 * sizeof(float), 256); }

// per-row-tile BatchNorm statistics written by the conv epilogue: [tiles][3][C_out]
inline size_t stats_tiles_bytes(const nsg_conv_desc *d)
{
    const int64_t out_pix = (int64_t)d->B * d->OH * d->OW;
    const int64_t tiles = d->transposed ? 4 * nsg_cdiv(out_pix / 4 + d->B * (d->OH + d->OW), 128) + 8 : nsg_cdiv(out_pix, 128);
    return nsg_align_up(nsg_bn_tiles_bytes(tiles + 8, d->C_out), 256);
}

GatherGemmParams gg_1x1(const void *in, const void *w, const float *bias, void *out, int64_t M, int CI, int CO, int flags, int in_dtype,
                        int out_dtype)
{
    GatherGemmParams p = {};
    p.in = in; p.w = w; p.bias = bias; p.out = out;
    p.in_dtype = in_dtype; p.out_dtype = out_dtype;
    p.B = 1; p.IH = 1; p.IW = (int)M; p.CI = CI;
    p.OH = 1; p.OW = (int)M; p.CO = CO;
    p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.pad_w = 0;
    p.mode = 0; p.M = (int)M; p.RH = 1; p.RW = (int)M;
    p.flags = flags;
    return p;
}

}  // namespace

extern "C" {

size_t nsg_packed_weight_floats(const nsg_conv_desc *d)
{
    if (!d) return 0;
    const size_t n = (size_t)kh_of(d) * kw_of(d) * d->C_in * d->C_out;
    // the single-channel layers keep one image as fp32 [C][16] for the stencil kernels whatever d->dtype is
    const bool c1 = (!d->transposed && d->C_in == 1) || (d->transposed && d->C_out == 1);
    // bf16 layers gemm_patch.hip can run carry a second, fragment-ordered image behind each plain one
    const bool frag = d->dtype == NSG_BF16 && (nsg_frag_image(d->C_out, d->C_in) || nsg_frag_image(d->C_in, d->C_out));
    return ((c1 && d->dtype == NSG_BF16) || frag) ? 2 * n : n;
}

int nsg_pack_conv_weights(const nsg_conv_desc *d, const float *w, void *w_fwd, void *w_dgrad, void *stream)
{
    return nsg_pack_conv_weights_batch(1, d, &w, &w_fwd, &w_dgrad, stream);
}

int nsg_pack_conv_weights_batch(int32_t n, const nsg_conv_desc *descs, const float *const *w, void *const *w_fwd,
                                void *const *w_dgrad, void *stream)
{
    NSG_REQUIRE(n >= 0 && (n == 0 || (descs && w && w_fwd && w_dgrad)), NSG_E_INVALID, "nsg_pack_conv_weights_batch: bad argument");
    hipStream_t s = (hipStream_t)stream;
    PackJobs jobs;
    int nj = 0;
    int64_t biggest = 0;
    auto flush = [&]() -> int {
        if (nj == 0) return NSG_OK;
        int64_t nb = nsg_cdiv(biggest, 256);
        if (nb > 256) nb = 256;
        hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)nb, (unsigned)nj), dim3(256), 0, s, jobs);
        nj = 0;
        biggest = 0;
        return nsg_check_launch("pack_w_kernel");
    };
    int rc = NSG_OK;
    auto add = [&](const float *src, void *dst, int T, int NN, int CC, int sn, int sc, int flip, int bf) {
        if (!dst || rc) return;
        if (nj == PACK_MAX_JOBS) rc = flush();
        PackJob &j = jobs.job[nj++];
        j.src = src; j.dst = dst; j.T = T; j.NN = NN; j.CC = CC; j.sn = sn; j.sc = sc; j.flip = flip; j.bf16 = bf;
        j.frag = (bf && T > 1 && nsg_frag_image(NN, CC)) ? 1 : 0;
        const int64_t total = (int64_t)T * NN * CC;
        if (total > biggest) biggest = total;
    };
    for (int i = 0; i < n; ++i) {
        const nsg_conv_desc *d = descs + i;
        const int kind = classify(d, "nsg_pack_conv_weights");
        if (kind < 0) return kind;
        NSG_REQUIRE(w[i], NSG_E_INVALID, "nsg_pack_conv_weights: null weights");
        const int T = kh_of(d) * kw_of(d), CI = d->C_in, CO = d->C_out;
        const int bf = d->dtype == NSG_BF16 ? 1 : 0;
        switch (kind) {
        case K_CONV:  // w[co][ci][t]
            add(w[i], w_fwd[i], T, CO, CI, CI * T, T, 0, bf);
            add(w[i], w_dgrad[i], T, CI, CO, T, CI * T, d->stride == 1 ? 1 : 0, bf);
            break;
        case K_CONVT:  // w[ci][co][t]
            add(w[i], w_fwd[i], T, CO, CI, T, CO * T, 0, bf);
            add(w[i], w_dgrad[i], T, CI, CO, CO * T, T, 0, bf);
            break;
        case K_CONV_C1:  // w[co][t] : forward is the stencil kernel (fp32 [co][16] = the parameter's own layout)
            add(w[i], w_fwd[i], 1, CO, 16, 16, 1, 0, 0);
            add(w[i], w_dgrad[i], 1, 16, CO, 1, 16, 0, bf);   // [n=t][c=co]
            break;
        case K_CONVT_C1:  // w[ci][t] : the data gradient is the stencil kernel (fp32 [ci][16])
            add(w[i], w_fwd[i], 1, 16, CI, 1, 16, 0, bf);     // [n=t][c=ci]
            add(w[i], w_dgrad[i], 1, CI, 16, 16, 1, 0, 0);
            break;
        }
        if (rc) return rc;
    }
    return flush();
}

size_t nsg_conv_workspace_bytes(const nsg_conv_desc *d)
{
    if (!d) return 0;
    const int kind = classify(d, "nsg_conv_workspace_bytes");
    if (kind < 0) return 0;
    const int T = kh_of(d) * kw_of(d);
    const int64_t Mp = lowres_pixels(d);
    size_t bytes = 0;
    int A, C, taps;
    if (kind == K_CONV) { A = d->C_out; C = d->C_in; taps = T; }
    else if (kind == K_CONVT) { A = d->C_in; C = d->C_out; taps = T; }
    else {   // single-channel layers: the dots / patches staging image of the GEMM-side passes, or the stencil wgrad's partials
        const size_t sb = nsg_c1_stencil_wgrad_workspace_bytes(kind == K_CONV_C1 ? d->C_out : d->C_in);
        bytes += patches_bytes(d) > sb ? patches_bytes(d) : sb;
        bytes += colsum_ws_bytes((int64_t)d->B * d->OH * d->OW, d->C_out);
        return bytes;
    }
    bytes += nsg_align_up(nsg_wgrad_workspace_bytes(Mp, taps, A, C), 256);
    bytes += colsum_ws_bytes((int64_t)d->B * d->OH * d->OW, d->C_out);
    bytes += stats_tiles_bytes(d);
    return bytes;
}

static int conv_forward_impl(const nsg_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, int32_t flags,
                             void *workspace, size_t workspace_bytes, void *stream, float *stats, int *stats_tiles)
{
    const int kind = classify(d, "nsg_conv_forward");
    if (kind < 0) return kind;
    NSG_REQUIRE(x && w_fwd && y, NSG_E_INVALID, "nsg_conv_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int out_dtype = (flags & NSG_OUT_F32) ? NSG_F32 : d->dtype;
    if (kind == K_CONV || kind == K_CONVT) {
        GatherGemmParams p = {};
        p.in = x; p.w = w_fwd; p.bias = bias; p.out = y;
        p.in_dtype = d->dtype; p.out_dtype = out_dtype;
        p.B = d->B; p.IH = d->IH; p.IW = d->IW; p.CI = d->C_in;
        p.OH = d->OH; p.OW = d->OW; p.CO = d->C_out;
        p.KH = kh_of(d); p.KW = kw_of(d); p.stride = d->stride; p.pad = ph_of(d); p.pad_w = pw_of(d);
        p.flags = flags & (NSG_RELU_IN | NSG_TANH_OUT | NSG_RELU_OUT);
        if (kind == K_CONV) { p.mode = 0; p.RH = d->OH; p.RW = d->OW; }
        else                { p.mode = 1; p.RH = d->IH; p.RW = d->IW; }
        p.M = d->B * p.RH * p.RW;
        p.stats = stats;
        p.stamps = g_debug_stamps;
        if (stats_tiles) *stats_tiles = nsg_gather_gemm_row_tiles(p);
        return nsg_launch_gather_gemm(p, s);
    }
    NSG_REQUIRE(workspace && workspace_bytes >= patches_bytes(d), NSG_E_WORKSPACE, "nsg_conv_forward: workspace too small");
    const int64_t Mp = lowres_pixels(d);
    if (kind == K_CONV_C1) {   // x is the fp32 single-channel image
        NSG_REQUIRE(!(flags & (NSG_RELU_IN | NSG_TANH_OUT | NSG_RELU_OUT)), NSG_E_UNSUPPORTED,
                    "nsg_conv_forward: no fused activations on the single-input-channel layer");
        NSG_REQUIRE(stats == nullptr, NSG_E_INVALID, "nsg_conv_forward: internal: tile statistics on the stencil path");
        NSG_REQUIRE(nsg_aligned16(w_fwd) && nsg_aligned16(y), NSG_E_INVALID, "nsg_conv_forward: operands must be 16-byte aligned");
        return nsg_launch_c1_stencil_fwd(reinterpret_cast<const float *>(x), reinterpret_cast<const float *>(w_fwd), bias, y, out_dtype,
                                         d->B, d->OH, d->OW, d->IH, d->IW, d->C_out, s);
    }
    NSG_REQUIRE(stats == nullptr, NSG_E_UNSUPPORTED, "nsg_conv_forward_bnstats: not available for a single-channel output");
    // K_CONVT_C1: per-input-pixel tap products (fp32), then the 4-tap gather with bias (+tanh) into the fp32 image y
    float *dots = reinterpret_cast<float *>(workspace);
    int rc = nsg_launch_gather_gemm(gg_1x1(x, w_fwd, nullptr, dots, Mp, d->C_in, 16, flags & NSG_RELU_IN, d->dtype, NSG_F32), s);
    if (rc) return rc;
    hipLaunchKernelGGL(col2im_c1_kernel<false>, dim3(col2im_blocks(d->B, d->OH, d->OW)), dim3(256), 0, s, dots, bias,
                       reinterpret_cast<float *>(y), d->B, d->IH, d->IW, d->OH, d->OW, (flags & NSG_TANH_OUT) ? 1 : 0, (const float *)nullptr, 0,
                       0.f, (float *)nullptr, (double *)nullptr);
    return nsg_check_launch("col2im_c1_kernel");
}

int nsg_conv_forward(const nsg_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, int32_t flags,
                     void *workspace, size_t workspace_bytes, void *stream)
{
    return conv_forward_impl(d, x, w_fwd, bias, y, flags, workspace, workspace_bytes, stream, nullptr, nullptr);
}

int nsg_conv_forward_bnstats(const nsg_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y,
                             int32_t flags, float eps, float momentum, float *mean, float *invstd, float *running_mean,
                             float *running_var, void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(d && mean && invstd, NSG_E_INVALID, "nsg_conv_forward_bnstats: null pointer");
    NSG_REQUIRE(!(flags & (NSG_TANH_OUT | NSG_RELU_OUT)), NSG_E_UNSUPPORTED, "nsg_conv_forward_bnstats: statistics are of the linear output");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_conv_workspace_bytes(d), NSG_E_WORKSPACE, "nsg_conv_forward_bnstats: workspace too small");
    if (!d->transposed && d->C_in == 1) {   // stencil layer: the statistics are a separate pass over y
        int rc = conv_forward_impl(d, x, w_fwd, bias, y, flags, workspace, workspace_bytes, stream, nullptr, nullptr);
        if (rc) return rc;
        const int ydt = (flags & NSG_OUT_F32) ? NSG_F32 : d->dtype;
        return nsg_bn_stats(y, (int64_t)d->B * d->OH * d->OW, d->C_out, ydt, eps, momentum, mean, invstd, running_mean, running_var,
                            workspace, workspace_bytes, stream);
    }
    // the tile statistics live at the END of the workspace (the C=1 staging image uses its start)
    const size_t tb = stats_tiles_bytes(d);
    float *tiles = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + nsg_conv_workspace_bytes(d) - tb);
    int ntiles = 0;
    int rc = conv_forward_impl(d, x, w_fwd, bias, y, flags, workspace, workspace_bytes, stream, tiles, &ntiles);
    if (rc) return rc;
    NSG_REQUIRE(nsg_bn_tiles_bytes(ntiles, d->C_out) <= tb, NSG_E_WORKSPACE, "nsg_conv_forward_bnstats: tile buffer too small");
    const int64_t M = (int64_t)d->B * d->OH * d->OW;
    return nsg_bn_stats_from_tiles(tiles, ntiles, M, d->C_out, eps, momentum, mean, invstd, running_mean, running_var,
                                   (hipStream_t)stream);
}

int nsg_conv_dgrad(const nsg_conv_desc *d, const void *dy, const void *w_dgrad, void *dx, int32_t flags, void *workspace,
                   size_t workspace_bytes, void *stream)
{
    return nsg_conv_dgrad_relu_add(d, dy, w_dgrad, nullptr, nullptr, dx, flags, workspace, workspace_bytes, stream);
}

int nsg_conv_dgrad_relu_add(const nsg_conv_desc *d, const void *dy, const void *w_dgrad, const void *add, const void *relu_x,
                            void *dx, int32_t flags, void *workspace, size_t workspace_bytes, void *stream)
{
    (void)flags;
    const int kind = classify(d, "nsg_conv_dgrad");
    if (kind < 0) return kind;
    NSG_REQUIRE(dy && w_dgrad && dx, NSG_E_INVALID, "nsg_conv_dgrad: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (kind == K_CONV || kind == K_CONVT) {
        // roles swap: the gradient of a conv is a transposed conv and vice versa
        GatherGemmParams p = {};
        p.in = dy; p.w = w_dgrad; p.bias = nullptr; p.out = dx;
        p.epi_add = add; p.epi_mask = relu_x;
        p.in_dtype = d->dtype; p.out_dtype = d->dtype;
        p.B = d->B; p.IH = d->OH; p.IW = d->OW; p.CI = d->C_out;
        p.OH = d->IH; p.OW = d->IW; p.CO = d->C_in;
        p.KH = kh_of(d); p.KW = kw_of(d);
        p.flags = 0;
        if (kind == K_CONV && d->stride == 1) {
            p.mode = 0; p.stride = 1; p.pad = kh_of(d) - 1 - ph_of(d); p.pad_w = kw_of(d) - 1 - pw_of(d); p.RH = d->IH; p.RW = d->IW;  // flipped taps
        } else if (kind == K_CONV) {
            p.mode = 1; p.stride = 2; p.pad = 1; p.pad_w = 1; p.RH = (d->IH + 1) / 2; p.RW = (d->IW + 1) / 2;  // 4/2/1: transposed classes
        } else {
            p.mode = 0; p.stride = 2; p.pad = 1; p.pad_w = 1; p.RH = d->IH; p.RW = d->IW;  // gradient of convT = strided conv
        }
        p.M = d->B * p.RH * p.RW;
        p.stamps = g_debug_stamps;
        return nsg_launch_gather_gemm(p, s);
    }
    NSG_REQUIRE(!add && !relu_x, NSG_E_UNSUPPORTED, "nsg_conv_dgrad_relu_add: not available for the single-channel layers");
    NSG_REQUIRE(workspace && workspace_bytes >= patches_bytes(d), NSG_E_WORKSPACE, "nsg_conv_dgrad: workspace too small");
    const int64_t Mp = lowres_pixels(d);
    if (kind == K_CONVT_C1) {
        // dy is the fp32 image: dx[pix][ci] = sum_t patch(dy)[pix][t] * w[ci][t]
        NSG_REQUIRE(nsg_aligned16(w_dgrad) && nsg_aligned16(dx), NSG_E_INVALID, "nsg_conv_dgrad: operands must be 16-byte aligned");
        return nsg_launch_c1_stencil_fwd(reinterpret_cast<const float *>(dy), reinterpret_cast<const float *>(w_dgrad), nullptr, dx,
                                         d->dtype, d->B, d->IH, d->IW, d->OH, d->OW, d->C_in, s);
    }
    // K_CONV_C1: dots[pix][t] = sum_co dy[pix][co] * w[co][t] (fp32), scattered back onto the fp32 image dx
    float *dots = reinterpret_cast<float *>(workspace);
    int rc = nsg_launch_gather_gemm(gg_1x1(dy, w_dgrad, nullptr, dots, Mp, d->C_out, 16, 0, d->dtype, NSG_F32), s);
    if (rc) return rc;
    hipLaunchKernelGGL(col2im_c1_kernel<false>, dim3(col2im_blocks(d->B, d->IH, d->IW)), dim3(256), 0, s, dots,
                       (const float *)nullptr, reinterpret_cast<float *>(dx), d->B, d->OH, d->OW, d->IH, d->IW, 0, (const float *)nullptr, 0, 0.f,
                       (float *)nullptr, (double *)nullptr);
    return nsg_check_launch("col2im_c1_kernel");
}

int nsg_conv_wgrad(const nsg_conv_desc *d, const void *x, const void *dy, float *dw, float *dbias, int32_t flags,
                   void *workspace, size_t workspace_bytes, void *stream)
{
    const int kind = classify(d, "nsg_conv_wgrad");
    if (kind < 0) return kind;
    NSG_REQUIRE(x && dy && dw, NSG_E_INVALID, "nsg_conv_wgrad: null pointer");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_conv_workspace_bytes(d), NSG_E_WORKSPACE, "nsg_conv_wgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace);
    const int64_t Mp = lowres_pixels(d);
    const int relu_x = (flags & NSG_RELU_IN) ? 1 : 0;
    if (kind == K_CONV_C1 || kind == K_CONVT_C1) {
        const size_t sb = nsg_c1_stencil_wgrad_workspace_bytes(kind == K_CONV_C1 ? d->C_out : d->C_in);
        NSG_REQUIRE(workspace_bytes >= sb + colsum_ws_bytes((int64_t)d->B * d->OH * d->OW, d->C_out), NSG_E_WORKSPACE,
                    "nsg_conv_wgrad: workspace too small");
        if (kind == K_CONV_C1) {   // x is the fp32 image, dy the C_out-channel tensor; dbias = its column sums
            NSG_REQUIRE(!relu_x, NSG_E_UNSUPPORTED, "nsg_conv_wgrad: NSG_RELU_IN on a single-channel input");
            NSG_REQUIRE(nsg_aligned16(dy), NSG_E_INVALID, "nsg_conv_wgrad: dy must be 16-byte aligned");
            return nsg_launch_c1_stencil_wgrad(reinterpret_cast<const float *>(x), dy, d->dtype, 0, dw, dbias, d->B, d->OH, d->OW, d->IH,
                                               d->IW, d->C_out, ws, sb, s);
        }
        // K_CONVT_C1: dy is the fp32 image, x the C_in-channel tensor; dbias = sum of the image
        NSG_REQUIRE(nsg_aligned16(x), NSG_E_INVALID, "nsg_conv_wgrad: x must be 16-byte aligned");
        int rc = nsg_launch_c1_stencil_wgrad(reinterpret_cast<const float *>(dy), x, d->dtype, relu_x, dw, nullptr, d->B, d->IH, d->IW,
                                             d->OH, d->OW, d->C_in, ws, sb, s);
        if (rc) return rc;
        if (dbias) return colsum(dy, NSG_F32, (int64_t)d->B * d->OH * d->OW, 1, dbias, ws + sb, s);
        return NSG_OK;
    }
    WgradParams p = {};
    p.dtype = d->dtype;
    p.B = d->B; p.KH = kh_of(d); p.KW = kw_of(d); p.stride = d->stride; p.pad = ph_of(d); p.pad_w = pw_of(d);
    p.Mp = (int)Mp;
    if (kind == K_CONV) {
        p.P = dy; p.PH = d->OH; p.PW = d->OW; p.A = d->C_out;
        p.Q = x; p.QH = d->IH; p.QW = d->IW; p.C = d->C_in; p.relu_q = relu_x;
    } else {   // K_CONVT
        p.P = x; p.PH = d->IH; p.PW = d->IW; p.A = d->C_in; p.relu_p = relu_x;
        p.Q = dy; p.QH = d->OH; p.QW = d->OW; p.C = d->C_out;
    }
    const size_t wg_bytes = nsg_align_up(nsg_wgrad_workspace_bytes(Mp, p.KH * p.KW, p.A, p.C), 256);
    int rc = nsg_launch_wgrad(p, dw, ws, wg_bytes, s);
    if (rc) return rc;
    ws += wg_bytes;
    if (dbias) return colsum(dy, d->dtype, (int64_t)d->B * d->OH * d->OW, d->C_out, dbias, ws, s);
    return NSG_OK;
}

/* ---- decoder.4 .. decoder.6 as one operator: BatchNorm2d -> ReLU -> ConvTranspose2d(C, 1, 4, 2, 1) [-> Tanh] ---- */
int32_t nsg_bn_relu_c1convt_supported(int32_t dtype, int32_t C) { return dtype == NSG_BF16 && nsg_c1m_supported(C) ? 1 : 0; }

size_t nsg_bn_relu_c1convt_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C)
{
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    const size_t fwd = nsg_align_up((size_t)B * H * W * 16 * sizeof(float), 256) + 2 * C1_LOSS_BLOCKS * sizeof(double);    // dots | loss, bias-gradient partials
    const size_t bwd = nsg_align_up((size_t)1024 * 2 * C * sizeof(float), 256) + nsg_c1_stencil_wgrad_workspace_bytes(C) +
                       colsum_ws_bytes((int64_t)B * 4 * H * W, 1);
    return fwd > bwd ? fwd : bwd;
}

int nsg_bn_relu_c1convt_forward(const void *u, int32_t dtype, const float *mean, const float *invstd, const float *gamma, const float *beta,
                                const float *w, const float *bias, float *y, int32_t flags, int32_t B, int32_t H, int32_t W, int32_t C,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(u && mean && invstd && gamma && beta && w && y, NSG_E_INVALID, "nsg_bn_relu_c1convt_forward: null pointer");
    NSG_REQUIRE(B > 0 && H > 0 && W > 0, NSG_E_INVALID, "nsg_bn_relu_c1convt_forward: bad extent");
    NSG_REQUIRE(nsg_bn_relu_c1convt_supported(dtype, C), NSG_E_UNSUPPORTED,
                "nsg_bn_relu_c1convt_forward: needs bf16 tensors and C = 32, 64, 96 or 128 (use the separate operators otherwise)");
    NSG_REQUIRE(!(flags & ~NSG_TANH_OUT), NSG_E_UNSUPPORTED, "nsg_bn_relu_c1convt_forward: only NSG_TANH_OUT");
    NSG_REQUIRE(nsg_aligned16(u) && nsg_aligned16(w), NSG_E_INVALID, "nsg_bn_relu_c1convt_forward: u and w must be 16-byte aligned");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_bn_relu_c1convt_workspace_bytes(B, H, W, C), NSG_E_WORKSPACE,
                "nsg_bn_relu_c1convt_forward: workspace too small");
    NSG_REQUIRE((int64_t)B * H * W * C <= 0x7fffffffLL * 4, NSG_E_UNSUPPORTED, "nsg_bn_relu_c1convt_forward: tensor too large");
    hipStream_t s = (hipStream_t)stream;
    float *dots = reinterpret_cast<float *>(workspace);
    int rc = nsg_launch_bnrelu_dots(u, mean, invstd, gamma, beta, w, dots, (int64_t)B * H * W, C, s);
    if (rc) return rc;
    hipLaunchKernelGGL(col2im_c1_kernel<false>, dim3(col2im_blocks(B, 2 * H, 2 * W)), dim3(256), 0, s, dots, bias, y, B, H, W, 2 * H,
                       2 * W, (flags & NSG_TANH_OUT) ? 1 : 0, (const float *)nullptr, 0, 0.f, (float *)nullptr, (double *)nullptr);
    return nsg_check_launch("col2im_c1_kernel");
}

/* nsg_bn_relu_c1convt_forward (with Tanh) + the reconstruction loss of train.py:118-129 in the same pass: loss_out[0] =
 * mean((pad(x_tilde) - target)^2) over the target's [B][2H][T] elements (T >= 2W: x_tilde zero-padded on the right),
 * dpre [B][2H][2W] = grad_scale * 2 / (B 2H T) * (x_tilde - target) * (1 - x_tilde^2): the gradient w.r.t. the Tanh's INPUT
 * (feed it to nsg_bn_relu_c1convt_backward); y = x_tilde is stored only when not NULL; dbias [1] or NULL = sum of dpre = the
 * transposed conv's bias gradient (then pass dbias = NULL to the backward: it would re-read dpre for it). */
int nsg_bn_relu_c1convt_forward_mse(const void *u, int32_t dtype, const float *mean, const float *invstd, const float *gamma, const float *beta,
                                    const float *w, const float *bias, float *y, const float *target, int32_t T, float grad_scale,
                                    float *loss_out, float *dpre, float *dbias, int32_t B, int32_t H, int32_t W, int32_t C, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(u && mean && invstd && gamma && beta && w && target && loss_out && dpre, NSG_E_INVALID, "nsg_bn_relu_c1convt_forward_mse: null pointer");
    NSG_REQUIRE(B > 0 && H > 0 && W > 0 && T >= 2 * W, NSG_E_INVALID, "nsg_bn_relu_c1convt_forward_mse: bad extent (T >= 2W)");
    NSG_REQUIRE(nsg_bn_relu_c1convt_supported(dtype, C), NSG_E_UNSUPPORTED,
                "nsg_bn_relu_c1convt_forward_mse: needs bf16 tensors and C = 32, 64, 96, 128 or 256 (use the separate operators otherwise)");
    NSG_REQUIRE(nsg_aligned16(u) && nsg_aligned16(w), NSG_E_INVALID, "nsg_bn_relu_c1convt_forward_mse: u and w must be 16-byte aligned");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_bn_relu_c1convt_workspace_bytes(B, H, W, C), NSG_E_WORKSPACE,
                "nsg_bn_relu_c1convt_forward_mse: workspace too small");
    NSG_REQUIRE((int64_t)B * H * W * C <= 0x7fffffffLL * 4, NSG_E_UNSUPPORTED, "nsg_bn_relu_c1convt_forward_mse: tensor too large");
    hipStream_t s = (hipStream_t)stream;
    float *dots = reinterpret_cast<float *>(workspace);
    double *partial = reinterpret_cast<double *>(reinterpret_cast<char *>(workspace) + nsg_align_up((size_t)B * H * W * 16 * sizeof(float), 256));
    int rc = nsg_launch_bnrelu_dots(u, mean, invstd, gamma, beta, w, dots, (int64_t)B * H * W, C, s);
    if (rc) return rc;
    const int64_t n = (int64_t)B * 2 * H * T;
    const int nb = col2im_blocks(B, 2 * H, 2 * W);
    hipLaunchKernelGGL(col2im_c1_kernel<true>, dim3(nb), dim3(256), 0, s, dots, bias, y, B, H, W, 2 * H, 2 * W, 1, target, T,
                       grad_scale * 2.0f / (float)n, dpre, partial);
    rc = nsg_check_launch("col2im_c1_kernel<loss>");
    if (rc) return rc;
    rc = nsg_launch_final_mean(partial, nb, (double)n, loss_out, s);
    if (rc || !dbias) return rc;
    return nsg_launch_final_mean(partial + nb, nb, 1.0, dbias, s);      // sum of dpre: the transposed conv's bias gradient
}

int nsg_bn_relu_c1convt_backward(const void *u, int32_t dtype, const float *mean, const float *invstd, const float *gamma, const float *beta,
                                 const float *w, const float *dy, void *du, float *du_colsum, float *dw, float *dbias, float *dgamma,
                                 float *dbeta, int32_t B, int32_t H, int32_t W, int32_t C, void *workspace, size_t workspace_bytes,
                                 void *stream)
{
    NSG_REQUIRE(u && mean && invstd && gamma && beta && w && dy && du && dw && dgamma && dbeta, NSG_E_INVALID,
                "nsg_bn_relu_c1convt_backward: null pointer");
    NSG_REQUIRE(B > 0 && H > 0 && W > 0, NSG_E_INVALID, "nsg_bn_relu_c1convt_backward: bad extent");
    NSG_REQUIRE(nsg_bn_relu_c1convt_supported(dtype, C), NSG_E_UNSUPPORTED,
                "nsg_bn_relu_c1convt_backward: needs bf16 tensors and C = 32, 64, 96 or 128 (use the separate operators otherwise)");
    NSG_REQUIRE(nsg_aligned16(u) && nsg_aligned16(du) && nsg_aligned16(w), NSG_E_INVALID, "nsg_bn_relu_c1convt_backward: u, du and w must be 16-byte aligned");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_bn_relu_c1convt_workspace_bytes(B, H, W, C), NSG_E_WORKSPACE,
                "nsg_bn_relu_c1convt_backward: workspace too small");
    NSG_REQUIRE((int64_t)B * H * W * C <= 0x7fffffffLL * 4, NSG_E_UNSUPPORTED, "nsg_bn_relu_c1convt_backward: tensor too large");
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace);
    float *sums = reinterpret_cast<float *>(ws);
    ws += nsg_align_up((size_t)1024 * 2 * C * sizeof(float), 256);
    float *partial17 = reinterpret_cast<float *>(ws);
    ws += nsg_c1_stencil_wgrad_workspace_bytes(C);
    const int64_t ntiles = (int64_t)B * H * ((W + 63) / 64);
    const int blocks = (int)(ntiles < 1024 ? ntiles : 1024);
    const float inv_m = 1.f / (float)((int64_t)B * H * W);
    int rc = nsg_launch_c1m_out_bwd_sums(dy, w, u, mean, invstd, gamma, beta, sums, blocks, B, H, W, C, s);
    if (rc) return rc;
    rc = nsg_launch_bn_bwd_final(sums, blocks, C, dgamma, dbeta, s);
    if (rc) return rc;
    rc = nsg_launch_c1m_out_bwd_apply(dy, w, u, mean, invstd, gamma, beta, dgamma, dbeta, inv_m, partial17, du, blocks, B, H, W, C, s);
    if (rc) return rc;
    rc = nsg_launch_c1_stencil_wgrad_final(partial17, blocks, C, dw, du_colsum, s);
    if (rc) return rc;
    if (dbias) return colsum(dy, NSG_F32, (int64_t)B * 4 * H * W, 1, dbias, ws, s);
    return NSG_OK;
}

/* ---- the ResBlock's 1x1 conv with the BatchNorm work around it folded into its operand staging (bf16, C = 32, 64, 128) ---- */
int32_t nsg_bn_relu_conv1x1_supported(int32_t dtype, int32_t C) { return nsg_flat1x1_supported(dtype, C) ? 1 : 0; }

size_t nsg_bn_relu_conv1x1_workspace_bytes(int64_t M, int32_t C)
{
    if (M <= 0 || C <= 0) return 0;
    const size_t wg = nsg_align_up(nsg_wgrad_workspace_bytes(M, 1, C, C), 256);
    const size_t fl = nsg_flat1x1_workspace_bytes(C);
    return wg > fl ? wg : fl;
}

namespace {
int check_1x1(const char *fn, int64_t M, int C, int dtype, size_t ws_bytes, const void *ws)
{
    if (!nsg_flat1x1_supported(dtype, C)) return nsg_fail(NSG_E_UNSUPPORTED, "%s: needs bf16 tensors and C = 32, 64, 128 or 256 (use the separate operators otherwise)", fn);
    if (M <= 0 || M >= 0x7fffffffLL / C) return nsg_fail(NSG_E_UNSUPPORTED, "%s: M = %lld rows not supported", fn, (long long)M);
    if (!ws || ws_bytes < nsg_bn_relu_conv1x1_workspace_bytes(M, C)) return nsg_fail(NSG_E_WORKSPACE, "%s: workspace too small", fn);
    return NSG_OK;
}
}  // namespace

int nsg_bn_relu_conv1x1_forward(const void *x, const float *mean, const float *invstd, const float *gamma, const float *beta, const float *w,
                                const float *bias, void *y, int64_t M, int32_t C, int32_t dtype, void *workspace, size_t workspace_bytes,
                                void *stream)
{
    NSG_REQUIRE(x && mean && invstd && gamma && beta && w && y, NSG_E_INVALID, "nsg_bn_relu_conv1x1_forward: null pointer");
    int rc = check_1x1("nsg_bn_relu_conv1x1_forward", M, C, dtype, workspace_bytes, workspace);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(x) && nsg_aligned16(y) && (!bias || nsg_aligned16(bias)), NSG_E_INVALID, "nsg_bn_relu_conv1x1_forward: pointers must be 16-byte aligned");
    return nsg_launch_flat1x1_forward(x, mean, invstd, gamma, beta, w, bias, y, M, C, workspace, 0, nullptr, (hipStream_t)stream);
}

/* The same with the batch statistics of y (the input of the BatchNorm that follows) taken from the kernel's store phase:
 * mean_y / invstd_y [C] out, running statistics updated (nn.BatchNorm2d training semantics), as nsg_bn_stats(y) would. */
int nsg_bn_relu_conv1x1_forward_bnstats(const void *x, const float *mean, const float *invstd, const float *gamma, const float *beta,
                                        const float *w, const float *bias, void *y, float eps, float momentum, float *mean_y,
                                        float *invstd_y, float *running_mean_y, float *running_var_y, int64_t M, int32_t C, int32_t dtype,
                                        void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && mean && invstd && gamma && beta && w && y && mean_y && invstd_y, NSG_E_INVALID, "nsg_bn_relu_conv1x1_forward_bnstats: null pointer");
    int rc = check_1x1("nsg_bn_relu_conv1x1_forward_bnstats", M, C, dtype, workspace_bytes, workspace);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(x) && nsg_aligned16(y) && (!bias || nsg_aligned16(bias)), NSG_E_INVALID,
                "nsg_bn_relu_conv1x1_forward_bnstats: pointers must be 16-byte aligned");
    int nblocks = 0;
    rc = nsg_launch_flat1x1_forward(x, mean, invstd, gamma, beta, w, bias, y, M, C, workspace, 1, &nblocks, (hipStream_t)stream);
    if (rc) return rc;
    return nsg_bn_stats_from_tiles(reinterpret_cast<const float *>(workspace), nblocks, M, C, eps, momentum, mean_y, invstd_y, running_mean_y,
                                   running_var_y, (hipStream_t)stream);
}

int nsg_bn_relu_conv1x1_wgrad(const void *x, const float *mean, const float *invstd, const float *gamma, const float *beta, const void *dy,
                              float *dw, int64_t M, int32_t C, int32_t dtype, void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(x && mean && invstd && gamma && beta && dy && dw, NSG_E_INVALID, "nsg_bn_relu_conv1x1_wgrad: null pointer");
    int rc = check_1x1("nsg_bn_relu_conv1x1_wgrad", M, C, dtype, workspace_bytes, workspace);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace);
    WgradParams p = {};
    p.dtype = NSG_BF16;
    p.B = 1; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.pad_w = 0;
    p.Mp = (int)M;
    p.P = dy; p.PH = 1; p.PW = (int)M; p.A = C;
    p.Q = x;  p.QH = 1; p.QW = (int)M; p.C = C;
    p.q_mean = mean; p.q_invstd = invstd; p.q_gamma = gamma; p.q_beta = beta;
    return nsg_launch_wgrad(p, dw, ws, nsg_align_up(nsg_wgrad_workspace_bytes(M, 1, C, C), 256), s);
}

int nsg_bn_backward_conv1x1_dgrad(const void *h, const void *dy, const float *mean, const float *invstd, const float *gamma,
                                  const float *dgamma, const float *dbeta, const float *w, void *dh, void *dx, float *dh_colsum,
                                  const void *prev_x, const float *prev_mean, const float *prev_invstd, const float *prev_gamma,
                                  const float *prev_beta, float *prev_dgamma, float *prev_dbeta, int64_t M, int32_t C, int32_t dtype,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(h && dy && mean && invstd && gamma && dgamma && dbeta && w && dh && dx, NSG_E_INVALID, "nsg_bn_backward_conv1x1_dgrad: null pointer");
    int rc = check_1x1("nsg_bn_backward_conv1x1_dgrad", M, C, dtype, workspace_bytes, workspace);
    if (rc) return rc;
    NSG_REQUIRE(nsg_aligned16(h) && nsg_aligned16(dy) && nsg_aligned16(dh) && nsg_aligned16(dx), NSG_E_INVALID,
                "nsg_bn_backward_conv1x1_dgrad: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (prev_x)
        NSG_REQUIRE(prev_mean && prev_invstd && prev_gamma && prev_beta && prev_dgamma && prev_dbeta && nsg_aligned16(prev_x), NSG_E_INVALID,
                    "nsg_bn_backward_conv1x1_dgrad: the BatchNorm in front needs all of its arguments (prev_x 16-byte aligned)");
    int nblocks = 0;
    float *prev_partial = nullptr;
    rc = nsg_launch_flat1x1_backward(h, dy, mean, invstd, gamma, dgamma, dbeta, w, dh, dx, M, C, workspace, &nblocks, prev_x, prev_mean,
                                     prev_invstd, prev_gamma, prev_beta, &prev_partial, s);
    if (rc) return rc;
    if (dh_colsum) {
        rc = nsg_launch_slab_sum_final(reinterpret_cast<const float *>(workspace), nblocks, C, dh_colsum, s);
        if (rc) return rc;
    }
    if (prev_x) return nsg_launch_bn_bwd_final(prev_partial, nblocks, C, prev_dgamma, prev_dbeta, s);
    return NSG_OK;
}

int32_t nsg_bn_backward_conv1x1_dgrad_wgrad_supported(int32_t dtype, int32_t C) { return nsg_flat1x1_fused_bwd_supported(dtype, C) ? 1 : 0; }

size_t nsg_bn_backward_conv1x1_dgrad_wgrad_workspace_bytes(int64_t M, int32_t C)
{
    if (M <= 0 || !nsg_flat1x1_fused_bwd_supported(NSG_BF16, C)) return 0;
    return nsg_flat1x1_fused_bwd_workspace_bytes(C);
}

int nsg_bn_backward_conv1x1_dgrad_wgrad(const void *h, const void *dy, const float *mean, const float *invstd, const float *gamma,
                                        const float *dgamma, const float *dbeta, const float *w, void *dx, float *dw, float *dh_colsum,
                                        const void *prev_x, const float *prev_mean, const float *prev_invstd, const float *prev_gamma,
                                        const float *prev_beta, float *prev_dgamma, float *prev_dbeta, int64_t M, int32_t C, int32_t dtype,
                                        void *workspace, size_t workspace_bytes, void *stream)
{
    const char *fn = "nsg_bn_backward_conv1x1_dgrad_wgrad";
    NSG_REQUIRE(h && dy && mean && invstd && gamma && dgamma && dbeta && w && dx && dw && prev_x && prev_mean && prev_invstd && prev_gamma &&
                prev_beta && prev_dgamma && prev_dbeta, NSG_E_INVALID, "nsg_bn_backward_conv1x1_dgrad_wgrad: null pointer");
    if (!nsg_flat1x1_fused_bwd_supported(dtype, C)) return nsg_fail(NSG_E_UNSUPPORTED, "%s: needs bf16 tensors and C = 128 (use nsg_bn_backward_conv1x1_dgrad + nsg_bn_relu_conv1x1_wgrad otherwise)", fn);
    if (M <= 0 || M >= 0x7fffffffLL / C) return nsg_fail(NSG_E_UNSUPPORTED, "%s: M = %lld rows not supported", fn, (long long)M);
    if (!workspace || workspace_bytes < nsg_flat1x1_fused_bwd_workspace_bytes(C)) return nsg_fail(NSG_E_WORKSPACE, "%s: workspace too small", fn);
    NSG_REQUIRE(nsg_aligned16(h) && nsg_aligned16(dy) && nsg_aligned16(dx) && nsg_aligned16(prev_x), NSG_E_INVALID,
                "nsg_bn_backward_conv1x1_dgrad_wgrad: tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int nblocks = 0;
    float *colsum_partial = nullptr, *prev_partial = nullptr, *dw_partial = nullptr;
    int rc = nsg_launch_flat1x1_fused_bwd(h, dy, mean, invstd, gamma, dgamma, dbeta, w, dx, dw, M, C, workspace, &nblocks, prev_x, prev_mean,
                                          prev_invstd, prev_gamma, prev_beta, &colsum_partial, &prev_partial, &dw_partial, s);
    if (rc) return rc;
    // one finaliser launch: the sums of the BatchNorm in front, the column sums of dh, the block partials of dw
    return nsg_launch_bn_bwd_final_wreduce(prev_partial, dh_colsum ? colsum_partial : nullptr, nblocks, C, prev_dgamma, prev_dbeta, dh_colsum,
                                           dw_partial, dw, C * C, s);
}

}  // extern "C"
