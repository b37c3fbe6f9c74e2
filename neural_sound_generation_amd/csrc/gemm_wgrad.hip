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
#include <type_traits>

// Diagnostics library only (-DNSG_DIAG): run-time switches; constants in libnsg.so
NSG_DIAG_SWITCH(unsigned long long *, g_wgrad_stamps, nullptr)
NSG_DIAG_SWITCH(int, g_wgrad_stagger, 0)
NSG_DIAG_SWITCH(int, g_wgrad_bf16_native, 1)   // 0: widen bf16 operands to fp32 at staging and use the fp32 MFMA (cross-check path)
#ifdef NSG_DIAG
static int g_wgrad_diag = 0;
extern "C" NSG_API void nsg_debug_set_wgrad_bf16_native(int on) { g_wgrad_bf16_native = on; }
extern "C" NSG_API void nsg_debug_set_wgrad_stagger(int units) { g_wgrad_stagger = units; }
extern "C" NSG_API void nsg_debug_set_wgrad_diag(int on) { g_wgrad_diag = on; }
extern "C" NSG_API void nsg_debug_set_wgrad_stamp_buffer(unsigned long long *buf) { g_wgrad_stamps = buf; }
#endif

namespace {

constexpr int KP = 32;  // pixels per chunk

template <typename TI, int WM, int WN, int TM, int TN, bool ONEHOT, bool DIAG = false>
__global__ __launch_bounds__(256) void wgrad_gemm_f32(const WgradParams p)
{
    constexpr int TA = WM * TM * 32;
    constexpr int TC = WN * TN * 32;
    constexpr int EPV = 16 / (int)sizeof(TI);            // elements per 16-byte piece: 4 (fp32) or 8 (bf16)
    constexpr int PA4 = TA / EPV, QC4 = TC / EPV;        // pieces per pixel row
    constexpr int PJ = (KP * PA4 + 255) / 256;           // pieces per thread per chunk
    constexpr int QJ = (KP * QC4 + 255) / 256;
    const TI *__restrict__ gP = reinterpret_cast<const TI *>(p.P);
    const TI *__restrict__ gQ = reinterpret_cast<const TI *>(p.Q);

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
    const float lbp = p.relu_p ? 0.f : -INFINITY;   // fused ReLU as a branch-free lower bound
    const float lbq = p.relu_q ? 0.f : -INFINITY;

    // two staging register sets: loads run TWO chunks ahead of the MFMAs (both operands stream from
    // HBM/MALL with no reuse inside a block, so one chunk of distance does not cover their latency)
    v4f rp[2][PJ], rq[2][QJ];
    int ridx[2][PJ];          // one-hot mode: the code of the row this thread stages
    unsigned okmask[2] = {0, 0};  // bit j: rp[j] real; bit 16+j: rq[j] real
    int g_mb = mbeg;          // first pixel of the NEXT chunk to load

    // Unconditional loads from clamped addresses; zero-fill / ReLU / one-hot expansion are done in
    // lstore() so no load is followed by a wait, and the steady-state loop is one basic block.
    auto gload = [&](auto set_c) {
        constexpr int S = decltype(set_c)::value;
        const int mb = g_mb;
        unsigned mk = 0;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int f = tid + 256 * j;
            const int pix = f / PA4;
            const int a4 = (f - pix * PA4) * EPV;
            const int m = mb + pix;
            const int ok = (pix < KP) & (m < mend) & ((a0 + a4) < p.A);
            if (ONEHOT) {
                ridx[S][j] = (int)p.idx[ok ? m : mbeg];
            } else {
                const size_t off = ok ? ((size_t)m * p.A + a0 + a4) : 0;
                rp[S][j] = *reinterpret_cast<const v4f *>(gP + off);
            }
            mk |= (unsigned)ok << j;
        }
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const int f = tid + 256 * j;
            const int pix = f / QC4;
            const int cc4 = (f - pix * QC4) * EPV;
            const int m = mb + pix;
            const int mm = m < mend ? m : mbeg;
            const int b = nsg_div(mm, p.div_phw);
            const int rem = mm - b * (p.PH * p.PW);
            const int py = nsg_div(rem, p.div_pw);
            const int px = rem - py * p.PW;
            const int qy = py * p.stride - p.pad + kh;
            const int qx = px * p.stride - p.pad_w + kw;
            const int ok = (pix < KP) & (m < mend) & ((c0 + cc4) < p.C) & (qy >= 0) & (qy < p.QH) & (qx >= 0) & (qx < p.QW);
            const size_t off = ok ? (((size_t)(b * p.QH + qy) * p.QW + qx) * p.C + c0 + cc4) : 0;
            rq[S][j] = *reinterpret_cast<const v4f *>(gQ + off);
            mk |= (unsigned)ok << (16 + j);
        }
        okmask[S] = mk;
        g_mb += KP;
    };
    auto lstore = [&](int buf, auto set_c) {
        constexpr int S = decltype(set_c)::value;
        float *ps = Ps + buf * KP * TA;
        float *qs = Qs + buf * KP * TC;
        // widen a staged 16-byte piece to EPV floats (fp32: as is; bf16: a 16-bit shift per element)
        auto widen = [&](const v4f &raw, float *o) {
            if constexpr (EPV == 4) {
                o[0] = raw.x; o[1] = raw.y; o[2] = raw.z; o[3] = raw.w;
            } else {
                const float r0 = raw.x, r1 = raw.y, r2 = raw.z, r3 = raw.w;
                const unsigned u[4] = {nsg_fbits(r0), nsg_fbits(r1), nsg_fbits(r2), nsg_fbits(r3)};
#pragma unroll
                for (int i = 0; i < 4; ++i) { o[2 * i] = nsg_bitsf(u[i] << 16); o[2 * i + 1] = nsg_bitsf(u[i] & 0xffff0000u); }
            }
        };
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int f = tid + 256 * j;
            if (f >= KP * PA4) continue;
            float o[EPV];
            const bool ok = (okmask[S] >> j) & 1u;
            if (ONEHOT) {
                const int a4 = (f - (f / PA4) * PA4) * EPV;
                const int code = ridx[S][j] - (a0 + a4);
#pragma unroll
                for (int e = 0; e < EPV; ++e) o[e] = (ok && code == e) ? 1.f : 0.f;
            } else {
                widen(rp[S][j], o);
#pragma unroll
                for (int e = 0; e < EPV; ++e) o[e] = ok ? fmaxf(o[e], lbp) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < EPV; e += 4) {
                v4f v = {o[e], o[e + 1], o[e + 2], o[e + 3]};
                *reinterpret_cast<v4f *>(ps + f * EPV + e) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const int f = tid + 256 * j;
            if (f >= KP * QC4) continue;
            float o[EPV];
            const bool ok = (okmask[S] >> (16 + j)) & 1u;
            widen(rq[S][j], o);
#pragma unroll
            for (int e = 0; e < EPV; ++e) o[e] = ok ? fmaxf(o[e], lbq) : 0.f;
#pragma unroll
            for (int e = 0; e < EPV; e += 4) {
                v4f v = {o[e], o[e + 1], o[e + 2], o[e + 3]};
                *reinterpret_cast<v4f *>(qs + f * EPV + e) = v;
            }
        }
    };

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Operands for 4 reduction steps (16 MFMAs) are fetched per batch into one of two register sets;
    // the NEXT batch's LDS reads are pinned into the middle of the current batch's MFMAs
    // (sched_group_barrier), so their latency is covered by >= 8 MFMAs (512 cycles) of queued work.
    // A wave cannot issue past an MFMA the matrix pipe has not accepted, so reads placed right in
    // front of their consumers expose ~60 cycles of LDS latency per 4-MFMA group (measured: 75 % of
    // the pipe instead of 95 %).
    auto compute = [&](int cur) {
        const float *ps = Ps + cur * KP * TA + wr * TM * 32 + l31 + h * TA;
        const float *qs = Qs + cur * KP * TC + wc * TN * 32 + l31 + h * TC;
        constexpr int NB = KP / 8;   // batches of 4 steps
        float a[2][4][TM], b[2][4][TN];
        auto fetch = [&](int kk, int set) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[set][u][i] = ps[2 * (4 * kk + u) * TA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[set][u][j] = qs[2 * (4 * kk + u) * TC + j * 32];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int kk = 0; kk < NB; ++kk) {
            const int set = kk & 1;
            if (kk + 1 < NB) fetch(kk + 1, set ^ 1);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][u][i], b[set][u][j], acc[i][j], 0, 0, 0);
            // order inside this batch: first half of the MFMAs, then the next batch's reads, then the rest
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);
            if (kk + 1 < NB) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);   // ds_read2_b32: 2 values per instruction
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);
        }
    };

    unsigned long long st_t0 = 0, st_r0 = 0;
    if (p.stamps) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    if (p.stagger > 0) {
        // Two blocks share each CU and run the same program: started together they want the matrix pipe
        // -- and then stage -- at the same times.  Delaying the later-dispatched block of each pair by a
        // fraction of a chunk puts one block's MFMA phase beside the other's staging phase.
        const int bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int nblk = gridDim.x * gridDim.y * gridDim.z;
        if (bid >= nblk / 2)
            for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    if (nchunk > 0) {
        // chunk c travels: gload -> register set (c & 1) -> lstore -> LDS buffer (c & 1) -> compute.
        // Iteration c: store chunk c+1 (loaded two iterations ago), re-issue that register set for
        // chunk c+3, then the MFMAs of chunk c.  Loads past the slab's end are clamped no-ops.
        gload(S0{});                 // chunk 0
        lstore(0, S0{});
        gload(S1{});                 // chunk 1
        gload(S0{});                 // chunk 2
        __syncthreads();
        int ch = 0;
        for (; ch + 2 < nchunk; ch += 2) {
            lstore(1, S1{});         // chunk ch+1
            gload(S1{});             // chunk ch+3
            compute(0);              // chunk ch
            __syncthreads();
            lstore(0, S0{});         // chunk ch+2
            gload(S0{});             // chunk ch+4
            compute(1);              // chunk ch+1
            __syncthreads();
        }
        if (ch + 1 < nchunk) {       // two chunks left: ch (in LDS 0) and ch+1 (in register set 1)
            lstore(1, S1{});
            compute(0);
            __syncthreads();
            compute(1);
        } else {                     // one chunk left, already in LDS 0
            compute(0);
        }
    }
    if (p.stamps && tid == 0) {   // diagnostics: never read by any kernel
        const int bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        p.stamps[2 * bid] = __builtin_amdgcn_s_memtime() - st_t0;
        p.stamps[2 * bid + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
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

// ------------------------------------------------------------------------------------------------
// bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  The reduction index (pixel) is the ROW
// of both NHWC operands, so both MFMA operands are needed transposed: lane (r, h) wants 8 consecutive
// pixels of ONE channel.  ds_read_b64_tr_b16 does that transpose in the LDS read path: per 16-lane
// group it reads a 4-row x 16-column block of 16-bit elements and hands lane i column i of the 4 rows.
// LDS tiles are plain [64 pixels][channels] bf16 copies of the rows with a pitch of 2*T + 64 bytes, so
// the 4 rows of a block land on 4 different 64-byte bank ranges (conflict-free).
// ------------------------------------------------------------------------------------------------
constexpr int KPB = 64;   // pixels per chunk (4 MFMA k-steps of 16)

__device__ __forceinline__ s16x4 lds_tr_read(const bf16_t *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
}

template <int WM, int WN, int TM, int TN, bool RELU, bool AFFQ = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wgrad_gemm_bf16(const WgradParams p)
{
    constexpr int TA = WM * TM * 32;
    constexpr int TC = WN * TN * 32;
    constexpr int PP = TA + 32;                          // LDS row pitch in bf16 elements (2*TA + 64 bytes)
    constexpr int QP = TC + 32;
    constexpr int PA8 = TA / 8, QC8 = TC / 8;            // 16-byte pieces per pixel row
    constexpr int PJ = KPB * PA8 / 256;
    constexpr int QJ = KPB * QC8 / 256;
    static_assert(KPB * PA8 % 256 == 0 && KPB * QC8 % 256 == 0, "every thread stages the same number of pieces");
    const bf16_t *__restrict__ gP = reinterpret_cast<const bf16_t *>(p.P);
    const bf16_t *__restrict__ gQ = reinterpret_cast<const bf16_t *>(p.Q);
    // buffer loads: an offset past the end reads as zero (slab tails, conv padding, ragged channel tiles)
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(gP), 0, (int)p.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(gQ), 0, (int)p.q_bytes, 0x00020000);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    bf16_t *Ps = reinterpret_cast<bf16_t *>(smem);       // [2][KPB][PP]
    bf16_t *Qs = Ps + 2 * KPB * PP;                      // [2][KPB][QP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int h = lane >> 5;
    // transposing read: lane = 16*g + 4*q + pp supplies the address of row q, columns 4pp..4pp+3 of its group's block
    const int g16 = (lane >> 4) & 1;                      // which 16-channel half of the 32-channel tile
    const int q = (lane >> 2) & 3, pp = lane & 3;

    const int slab = blockIdx.x;
    const int tap = blockIdx.y;
    const int ctiles = (p.C + TC - 1) / TC;
    const int a0 = (blockIdx.z / ctiles) * TA;
    const int c0 = (blockIdx.z % ctiles) * TC;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    const int mbeg = slab * p.slab_rows;
    const int mend = min(p.Mp, mbeg + p.slab_rows);
    const int nchunk = (mend - mbeg + KPB - 1) / KPB;

    // this thread's pieces: piece j of P is (pixel ppix + pstep*j, channels pa8..pa8+7); likewise Q
    constexpr int PSTEP = 256 / PA8, QSTEP = 256 / QC8;
    const int ppix = tid / PA8, pa8 = (tid % PA8) * 8;
    const int qpix = tid / QC8, qc8 = (tid % QC8) * 8;
    const bool pa_ok = (a0 + pa8) < p.A, qc_ok = (c0 + qc8) < p.C;
    const int yoff = kh - p.pad, xoff = kw - p.pad_w;

    v4f rp0[PJ], rq0[QJ], rp1[PJ], rq1[QJ];   // two register stages (loads run two chunks ahead of the MFMAs)
    int g_mb = mbeg;

    auto gload = [&](v4f (&rp)[PJ], v4f (&rq)[QJ]) {
        const int mb = g_mb;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int m = mb + ppix + PSTEP * j;
            const bool ok = pa_ok & (m < mend);
            const unsigned off = ok ? (unsigned)(m * p.A + a0 + pa8) * 2u : OOB;
            rp[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_p, (int)off, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const int m = mb + qpix + QSTEP * j;
            const int mm = m < mend ? m : mbeg;
            const int b = nsg_div(mm, p.div_phw);
            const int rem = mm - b * (p.PH * p.PW);
            const int py = nsg_div(rem, p.div_pw);
            const int px = rem - py * p.PW;
            const int qy = py * p.stride + yoff;
            const int qx = px * p.stride + xoff;
            const bool ok = qc_ok & (m < mend) & ((unsigned)qy < (unsigned)p.QH) & ((unsigned)qx < (unsigned)p.QW);
            const unsigned off = ok ? (unsigned)(((b * p.QH + qy) * p.QW + qx) * p.C + c0 + qc8) * 2u : OOB;
            rq[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_q, (int)off, 0, 0));
        }
        g_mb += KPB;
    };
    // bf16 ReLU: a bf16 is negative exactly when its bit pattern is a negative int16; floor = 0 (ReLU) or INT16_MIN (identity)
    const short pfl = p.relu_p ? (short)0 : (short)-32768, qfl = p.relu_q ? (short)0 : (short)-32768;
    auto floor8 = [](v4f v, short fl) {
        s16x8 sv = __builtin_bit_cast(s16x8, v);
        const s16x8 f8 = {fl, fl, fl, fl, fl, fl, fl, fl};
        return __builtin_bit_cast(v4f, __builtin_elementwise_max(sv, f8));
    };
    // AFFQ: this thread's 8 channels of Q (the same for every piece) go through max(fma(q, fs, off), 0): the BatchNorm + ReLU
    // that produced the conv's input, applied here instead of in a pass of its own (1x1 conv: no padding, and rows past
    // the slab multiply zero rows of P)
    float qfs[8], qoff[8];
    if (AFFQ) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + qc8 + e;
            qfs[e] = c < p.C ? p.q_invstd[c] * p.q_gamma[c] : 0.f;
            qoff[e] = c < p.C ? __builtin_fmaf(-p.q_mean[c], qfs[e], p.q_beta[c]) : 0.f;
        }
    }
    auto affq = [&](v4f v) {
        float x[8];
        Elem<bf16_t>::unpack16(v, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = fmaxf(__builtin_fmaf(x[e], qfs[e], qoff[e]), 0.f);
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f2_t __attribute__((ext_vector_type(2)));
        unsigned u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const f2_t f = {x[2 * i], x[2 * i + 1]}; u[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t)); }
        return v4f{nsg_bitsf(u[0]), nsg_bitsf(u[1]), nsg_bitsf(u[2]), nsg_bitsf(u[3])};
    };
    auto lstore = [&](int buf, const v4f (&rp)[PJ], const v4f (&rq)[QJ]) {
        bf16_t *ps = Ps + buf * KPB * PP;
        bf16_t *qs = Qs + buf * KPB * QP;
#pragma unroll
        for (int j = 0; j < PJ; ++j)
            *reinterpret_cast<v4f *>(ps + (ppix + PSTEP * j) * PP + pa8) = RELU ? floor8(rp[j], pfl) : rp[j];
#pragma unroll
        for (int j = 0; j < QJ; ++j)
            *reinterpret_cast<v4f *>(qs + (qpix + QSTEP * j) * QP + qc8) = AFFQ ? affq(rq[j]) : (RELU ? floor8(rq[j], qfl) : rq[j]);
    };

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int cur) {
        // this lane's address inside a (k-step, tile) block: row = 8h + q (+4 for the second half), column = 16*g16 + 4*pp
        const bf16_t *pbase = Ps + cur * KPB * PP + (8 * h + q) * PP + wr * TM * 32 + 16 * g16 + 4 * pp;
        const bf16_t *qbase = Qs + cur * KPB * QP + (8 * h + q) * QP + wc * TN * 32 + 16 * g16 + 4 * pp;
#pragma unroll
        for (int ks = 0; ks < KPB / 16; ++ks) {
            s16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const s16x4 lo = lds_tr_read(pbase + (16 * ks) * PP + i * 32);
                const s16x4 hi = lds_tr_read(pbase + (16 * ks + 4) * PP + i * 32);
                a[i] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const s16x4 lo = lds_tr_read(qbase + (16 * ks) * QP + j * 32);
                const s16x4 hi = lds_tr_read(qbase + (16 * ks + 4) * QP + j * 32);
                b[j] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                                       acc[i][j], 0, 0, 0);
        }
    };

    // chunk i computed from LDS buffer i&1, chunk i+1 waiting in registers, chunk i+2 in flight (see gather_gemm)
    if (nchunk > 0) {
        gload(rp0, rq0);
        if (nchunk > 1) gload(rp1, rq1);
        lstore(0, rp0, rq0);
        __syncthreads();
        int ch = 0;
        for (; ch + 2 < nchunk; ch += 2) {
            gload(rp0, rq0);          // chunk ch+2
            compute(0);
            lstore(1, rp1, rq1);
            __syncthreads();
            gload(rp1, rq1);          // chunk ch+3 (past the slab: every row fails m < mend, reads zeros, never stored)
            compute(1);
            lstore(0, rp0, rq0);
            __syncthreads();
        }
        compute(0);
        if (ch + 1 < nchunk) {
            lstore(1, rp1, rq1);
            __syncthreads();
            compute(1);
        }
    }

    const int ntaps = p.KH * p.KW;
    const int l31 = lane & 31;
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

// ------------------------------------------------------------------------------------------------
// index_add_ for the bf16 compute mode: dst[k][c] = sum over pixels with idx[pix] == k of g[pix][c]  (g fp32), as the
// same pixel-reduction GEMM with a generated one-hot P operand (exact in bf16) and g split into bf16 hi + lo parts
// (g = hi + lo + O(2^-17 |g|)): two v_mfma_f32_32x32x16_bf16 per k-step instead of eight fp32 MFMAs.  Deterministic
// (fixed slab order) like every other reduction here; relative error of a sum ~2^-17, far below bf16 storage.
// ------------------------------------------------------------------------------------------------
constexpr int KPO = 32;   // pixels per chunk (LDS: 3 tiles x 2 buffers must leave room for two workgroups per CU)

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void onehot_gemm_bf16x2(const WgradParams p)
{
    constexpr int TA = WM * TM * 32;
    constexpr int TC = WN * TN * 32;
    constexpr int PP = TA + 32, QP = TC + 32;            // LDS row pitches in bf16 elements
    constexpr int PA8 = TA / 8, QC8 = TC / 8;
    constexpr int PJ = KPO * PA8 / 256, QJ = KPO * QC8 / 256;
    static_assert(KPO * PA8 % 256 == 0 && KPO * QC8 % 256 == 0, "every thread stages the same number of pieces");
    const float *__restrict__ gQ = reinterpret_cast<const float *>(p.Q);
    const int64_t *__restrict__ gidx = p.idx;
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gQ), 0, (int)p.q_bytes, 0x00020000);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    bf16_t *Ps = reinterpret_cast<bf16_t *>(smem);       // [2][KPO][PP]
    bf16_t *Qh = Ps + 2 * KPO * PP;                      // [2][KPO][QP]
    bf16_t *Ql = Qh + 2 * KPO * QP;                      // [2][KPO][QP]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int h = lane >> 5;
    const int g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;

    const int slab = blockIdx.x;
    const int ctiles = (p.C + TC - 1) / TC;
    const int a0 = (blockIdx.z / ctiles) * TA;
    const int c0 = (blockIdx.z % ctiles) * TC;
    const int mbeg = slab * p.slab_rows;
    const int mend = min(p.Mp, mbeg + p.slab_rows);
    const int nchunk = (mend - mbeg + KPO - 1) / KPO;

    constexpr int PSTEP = 256 / PA8, QSTEP = 256 / QC8;
    const int ppix = tid / PA8, pa8 = (tid % PA8) * 8;
    const int qpix = tid / QC8, qc8 = (tid % QC8) * 8;
    const bool qc_ok = (c0 + qc8) < p.C;                 // C % 8 == 0: a piece is inside or outside as a whole

    int ri0[PJ], ri1[PJ];                                // code of each staged pixel (two register stages)
    v4f rq0[2 * QJ], rq1[2 * QJ];
    int g_mb = mbeg;
    auto gload = [&](int (&ri)[PJ], v4f (&rq)[2 * QJ]) {
        const int mb = g_mb;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int m = mb + ppix + PSTEP * j;
            const int64_t code = gidx[m < mend ? m : mbeg];      // clamped, unconditional
            ri[j] = m < mend ? (int)code : -0x40000000;          // a pixel past the slab matches no code
        }
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const int m = mb + qpix + QSTEP * j;
            const bool ok = qc_ok & (m < mend);
            const unsigned off = ok ? (unsigned)(m * p.C + c0 + qc8) * 4u : OOB;
            rq[2 * j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_q, (int)off, 0, 0));
            rq[2 * j + 1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_q, (int)(ok ? off + 16u : OOB), 0, 0));
        }
        g_mb += KPO;
    };
    auto lstore = [&](int buf, const int (&ri)[PJ], const v4f (&rq)[2 * QJ]) {
        bf16_t *ps = Ps + buf * KPO * PP;
        bf16_t *qh = Qh + buf * KPO * QP;
        bf16_t *ql = Ql + buf * KPO * QP;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const unsigned k = (unsigned)(ri[j] - (a0 + pa8));   // < 8 exactly when the pixel's code is one of this piece's 8
            const unsigned one = k < 8u ? (0x3f80u << (16 * (k & 1u))) : 0u;   // bf16 1.0 in the element's half of its dword
            const unsigned w = k >> 1;
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u v = {w == 0 ? one : 0u, w == 1 ? one : 0u, w == 2 ? one : 0u, w == 3 ? one : 0u};
            *reinterpret_cast<v4u *>(ps + (ppix + PSTEP * j) * PP + pa8) = v;
        }
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const float v[8] = {rq[2 * j].x, rq[2 * j].y, rq[2 * j].z, rq[2 * j].w, rq[2 * j + 1].x, rq[2 * j + 1].y, rq[2 * j + 1].z, rq[2 * j + 1].w};
            s16x8 hv, lv;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bf16_t hb = nsg_f2bf(v[e]);
                hv[e] = (short)hb;
                lv[e] = (short)nsg_f2bf(v[e] - nsg_bf2f(hb));
            }
            *reinterpret_cast<s16x8 *>(qh + (qpix + QSTEP * j) * QP + qc8) = hv;
            *reinterpret_cast<s16x8 *>(ql + (qpix + QSTEP * j) * QP + qc8) = lv;
        }
    };

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int cur) {
        const bf16_t *pbase = Ps + cur * KPO * PP + (8 * h + q) * PP + wr * TM * 32 + 16 * g16 + 4 * pp;
        const bf16_t *hbase = Qh + cur * KPO * QP + (8 * h + q) * QP + wc * TN * 32 + 16 * g16 + 4 * pp;
        const bf16_t *lbase = Ql + cur * KPO * QP + (8 * h + q) * QP + wc * TN * 32 + 16 * g16 + 4 * pp;
#pragma unroll
        for (int ks = 0; ks < KPO / 16; ++ks) {
            s16x8 a[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const s16x4 lo = lds_tr_read(pbase + (16 * ks) * PP + i * 32);
                const s16x4 hi = lds_tr_read(pbase + (16 * ks + 4) * PP + i * 32);
                a[i] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const s16x4 lo = lds_tr_read(hbase + (16 * ks) * QP + j * 32);
                const s16x4 hi = lds_tr_read(hbase + (16 * ks + 4) * QP + j * 32);
                bh[j] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const s16x4 lo2 = lds_tr_read(lbase + (16 * ks) * QP + j * 32);
                const s16x4 hi2 = lds_tr_read(lbase + (16 * ks + 4) * QP + j * 32);
                bl[j] = s16x8{lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, bl[j]), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
                }
        }
    };

    if (nchunk > 0) {
        gload(ri0, rq0);
        if (nchunk > 1) gload(ri1, rq1);
        lstore(0, ri0, rq0);
        __syncthreads();
        int ch = 0;
        for (; ch + 2 < nchunk; ch += 2) {
            gload(ri0, rq0);
            compute(0);
            lstore(1, ri1, rq1);
            __syncthreads();
            gload(ri1, rq1);
            compute(1);
            lstore(0, ri0, rq0);
            __syncthreads();
        }
        compute(0);
        if (ch + 1 < nchunk) {
            lstore(1, ri1, rq1);
            __syncthreads();
            compute(1);
        }
    }

    const int l31 = lane & 31;
    float *dst = p.partial + ((size_t)slab * p.A) * p.C;
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
// One thread per output when there are few slabs; otherwise 8 lanes per output split the slabs and
// the partial sums are added in lane order (fixed order either way: bitwise reproducible).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ dst, int nslab, int ntaps,
                                                           int A, int C, int split)
{
    __shared__ float red[256];
    const int64_t total = (int64_t)ntaps * A * C;
    const int per_block = 256 / split;
    const int tid = threadIdx.x;
    const int sub = tid / per_block;          // which share of the slabs
    const int loc = tid - sub * per_block;    // which output inside the block (consecutive -> coalesced)
    for (int64_t e0 = (int64_t)blockIdx.x * per_block; e0 < total; e0 += (int64_t)gridDim.x * per_block) {
        const int64_t e = e0 + loc;
        float sacc = 0.f;
        if (e < total) {
            const int chunk = (nslab + split - 1) / split;
            const int s0 = sub * chunk, s1 = min(nslab, s0 + chunk);
            if (s1 > s0) sacc = nsg_strided_sum<float>(partial + (size_t)s0 * total + e, (size_t)total, s1 - s0);
        }
        if (split > 1) {
            red[tid] = sacc;
            __syncthreads();
            if (sub == 0) {
                sacc = 0.f;
                for (int k = 0; k < split; ++k) sacc += red[k * per_block + loc];
            }
            __syncthreads();
        }
        if (sub == 0 && e < total) {
            const int c = (int)(e % C);
            const int a = (int)((e / C) % A);
            const int t = (int)(e / ((int64_t)A * C));
            dst[((size_t)a * C + c) * ntaps + t] = sacc;
        }
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
    int64_t want = 512 / tiles;                     // ONE full round of the 512 resident blocks (2 per CU): no stragglers, least slab traffic
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

template <typename TI, int WM, int WN, int TM, int TN, bool ONEHOT>
int launch_wg1(const WgradParams &p, int nslab, hipStream_t s)
{
    constexpr int TA = WM * TM * 32, TC = WN * TN * 32;
    const size_t lds = (size_t)2 * KP * (TA + TC) * sizeof(float);
    const int ntaps = p.KH * p.KW;
    const int tiles = (int)(nsg_cdiv(p.A, TA) * nsg_cdiv(p.C, TC));
    dim3 grid(nslab, ntaps, tiles);
    static LdsOptIn once;
    if (lds > 65536) {
        const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&wgrad_gemm_f32<TI, WM, WN, TM, TN, ONEHOT>)}, lds, "wgrad");
        if (rc != NSG_OK) return rc;
    }
    hipLaunchKernelGGL((wgrad_gemm_f32<TI, WM, WN, TM, TN, ONEHOT>), grid, dim3(256), lds, s, p);
    return nsg_check_launch("wgrad_gemm_f32");
}

template <int WM, int WN, int TM, int TN>
int launch_wg_bf16(const WgradParams &p, int nslab, hipStream_t s)
{
    constexpr int TA = WM * TM * 32, TC = WN * TN * 32;
    const size_t lds = (size_t)2 * KPB * ((TA + 32) + (TC + 32)) * sizeof(bf16_t);
    const int ntaps = p.KH * p.KW;
    const int tiles = (int)(nsg_cdiv(p.A, TA) * nsg_cdiv(p.C, TC));
    dim3 grid(nslab, ntaps, tiles);
    static LdsOptIn once;
    if (lds > 65536) {
        const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&wgrad_gemm_bf16<WM, WN, TM, TN, false>),
                                             reinterpret_cast<const void *>(&wgrad_gemm_bf16<WM, WN, TM, TN, true>),
                                             reinterpret_cast<const void *>(&wgrad_gemm_bf16<WM, WN, TM, TN, false, true>)}, lds, "wgrad");
        if (rc != NSG_OK) return rc;
    }
    if (p.q_mean) {
        if (p.relu_p || p.relu_q || p.KH * p.KW != 1 || p.pad || p.pad_w || p.stride != 1)
            return nsg_fail(NSG_E_UNSUPPORTED, "wgrad: the BatchNorm-on-load operand is for plain 1x1 convolutions");
        hipLaunchKernelGGL((wgrad_gemm_bf16<WM, WN, TM, TN, false, true>), grid, dim3(256), lds, s, p);
    } else if (p.relu_p || p.relu_q) hipLaunchKernelGGL((wgrad_gemm_bf16<WM, WN, TM, TN, true>), grid, dim3(256), lds, s, p);
    else                      hipLaunchKernelGGL((wgrad_gemm_bf16<WM, WN, TM, TN, false>), grid, dim3(256), lds, s, p);
    return nsg_check_launch("wgrad_gemm_bf16");
}

template <int WM, int WN, int TM, int TN>
int launch_wg_diag(const WgradParams &p, int nslab, hipStream_t s)
{
    constexpr int TA = WM * TM * 32, TC = WN * TN * 32;
    const size_t lds = (size_t)2 * KP * (TA + TC) * sizeof(float);
    dim3 grid(nslab, p.KH * p.KW, (unsigned)(nsg_cdiv(p.A, TA) * nsg_cdiv(p.C, TC)));
    hipLaunchKernelGGL((wgrad_gemm_f32<float, WM, WN, TM, TN, false, true>), grid, dim3(256), lds, s, p);
    return nsg_check_launch("wgrad_gemm_f32<diag>");
}

template <int WM, int WN, int TM, int TN>
int launch_onehot_bf16x2(const WgradParams &p, int nslab, hipStream_t s)
{
    constexpr int TA = WM * TM * 32, TC = WN * TN * 32;
    const size_t lds = (size_t)2 * KPO * ((TA + 32) + 2 * (TC + 32)) * sizeof(bf16_t);
    dim3 grid(nslab, 1, (unsigned)(nsg_cdiv(p.A, TA) * nsg_cdiv(p.C, TC)));
    hipLaunchKernelGGL((onehot_gemm_bf16x2<WM, WN, TM, TN>), grid, dim3(256), lds, s, p);
    return nsg_check_launch("onehot_gemm_bf16x2");
}

template <int WM, int WN, int TM, int TN>
int launch_wg(const WgradParams &p, int nslab, hipStream_t s)
{
    if constexpr (WN * TN * 32 >= 64) {
        if (p.onehot == 2) return launch_onehot_bf16x2<WM, WN, TM, TN>(p, nslab, s);   // one-hot on the bf16 pipe, split rows
    }
    if (p.onehot) return launch_wg1<float, WM, WN, TM, TN, true>(p, nslab, s);   // one-hot: Q (the scattered rows) is fp32
    if (p.dtype == NSG_BF16) return g_wgrad_bf16_native ? launch_wg_bf16<WM, WN, TM, TN>(p, nslab, s) : launch_wg1<bf16_t, WM, WN, TM, TN, false>(p, nslab, s);
    return launch_wg1<float, WM, WN, TM, TN, false>(p, nslab, s);
}

}  // namespace

// gemm_wgrad_strip.hip: the row-strip kernel for the bf16 3x3/1 and 4x4/2 layers with channels in multiples of 128
int nsg_wgrad_strip_slabs(int ntaps, int A, int C);
bool nsg_wgrad_strip_applicable(const WgradParams &p);
int nsg_launch_wgrad_strip(const WgradParams &p, int *nslab, hipStream_t s);

// dst[(a*C + c)*ntaps + t] = sum over slabs (slab order) of partial[slab][t][a][c]: the closing pass of every weight gradient,
// also used by gemm_flat.hip's fused data + weight gradient of the 1x1 conv
int nsg_launch_wgrad_reduce(const float *partial, float *dst, int nslab, int ntaps, int A, int C, hipStream_t s)
{
    const int64_t total = (int64_t)ntaps * A * C;
    const int split = (nslab >= 64 && total < 512 * 256) ? 8 : 1;   // few outputs, many slabs: share the slabs
    const int64_t nb = nsg_cdiv(total, 256 / split);
    const int blocks = (int)(nb > 4096 ? 4096 : nb);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, partial, dst, nslab, ntaps, A, C, split);
    return nsg_check_launch("wgrad_reduce_kernel");
}

size_t nsg_wgrad_workspace_bytes(int64_t Mp, int ntaps, int A, int C)
{
    const SlabPlan sp = plan_slabs(Mp, ntaps, A, C);
    int nslab = sp.nslab;
    if ((ntaps == 9 || ntaps == 16) && A % 128 == 0 && C % 128 == 0 && nsg_wgrad_strip_slabs(ntaps, A, C) > nslab)
        nslab = nsg_wgrad_strip_slabs(ntaps, A, C);       // whichever kernel runs, its slabs fit
    return (size_t)nslab * ntaps * A * C * sizeof(float);
}

int nsg_launch_wgrad(WgradParams p, float *dst, void *ws, size_t ws_bytes, hipStream_t s)
{
    const int ntaps = p.KH * p.KW;
    if (p.Mp <= 0) return nsg_fail(NSG_E_INVALID, "wgrad: empty reduction");
    // one-hot rows are generated, not loaded: any number of codes works there
    const int epv = (!p.onehot && p.dtype == NSG_BF16) ? 8 : 4;
    if ((!p.onehot && p.A % epv != 0) || p.C % epv != 0) return nsg_fail(NSG_E_UNSUPPORTED, "wgrad: channels (%d,%d) must be multiples of %d", p.A, p.C, epv);
    if ((!p.onehot && !nsg_aligned16(p.P)) || !nsg_aligned16(p.Q)) return nsg_fail(NSG_E_INVALID, "wgrad: operands must be 16-byte aligned");
    SlabPlan sp = plan_slabs(p.Mp, ntaps, p.A, p.C);
    const size_t need = nsg_wgrad_workspace_bytes(p.Mp, ntaps, p.A, p.C);
    if (ws == nullptr || ws_bytes < need) return nsg_fail(NSG_E_WORKSPACE, "wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    p.partial = reinterpret_cast<float *>(ws);
    p.slab_rows = sp.slab_rows;
    p.stamps = g_wgrad_stamps;
    p.stagger = g_wgrad_stagger;
    if (p.onehot == 2) {
        const uint64_t qb = (uint64_t)p.Mp * p.C * 4;
        if (qb >= 0xfffffff0ull) return nsg_fail(NSG_E_UNSUPPORTED, "index_add (bf16 pipe): operand larger than 4 GiB");
        if (p.C % 8 != 0) return nsg_fail(NSG_E_UNSUPPORTED, "index_add (bf16 pipe): D=%d must be a multiple of 8", p.C);
        p.q_bytes = (unsigned)qb;
    }
    if (!p.onehot && p.dtype == NSG_BF16) {   // 32-bit byte offsets in the bf16 kernel's buffer loads
        const uint64_t pb = (uint64_t)p.Mp * p.A * 2, qb = (uint64_t)p.B * p.QH * p.QW * p.C * 2;
        if (pb >= 0xfffffff0ull || qb >= 0xfffffff0ull) return nsg_fail(NSG_E_UNSUPPORTED, "wgrad: operand larger than 4 GiB: split the batch");
        p.p_bytes = (unsigned)pb;
        p.q_bytes = (unsigned)qb;
    }
    p.div_pw = nsg_fastdiv((uint32_t)p.PW);
    p.div_phw = nsg_fastdiv((uint32_t)p.PH * (uint32_t)p.PW);
    int rc;
    if (nsg_wgrad_strip_applicable(p)) {
        rc = nsg_launch_wgrad_strip(p, &sp.nslab, s);        // one kernel row per workgroup, strips of 64 pixels (gemm_wgrad_strip.hip)
    } else if (p.C <= 32) {
        rc = launch_wg<4, 1, 1, 1>(p, sp.nslab, s);          // 128 x 32 (im2col'd single-channel layers)
    } else if (p.A <= 64 && p.C <= 64) {
        rc = launch_wg<2, 2, 1, 1>(p, sp.nslab, s);          // 64 x 64
#ifdef NSG_DIAG
    } else if (g_wgrad_diag && p.stamps && !p.onehot) {
        rc = launch_wg_diag<2, 2, 2, 2>(p, sp.nslab, s);     // diagnostics build of the 128 x 128 kernel
#endif
    } else {
        rc = launch_wg<2, 2, 2, 2>(p, sp.nslab, s);          // 128 x 128
    }
    if (rc != NSG_OK) return rc;
    const int64_t total = (int64_t)ntaps * p.A * p.C;
    const int split = (sp.nslab >= 64 && total < 512 * 256) ? 8 : 1;   // few outputs, many slabs: share the slabs
    const int64_t nb = nsg_cdiv(total, 256 / split);
    const int blocks = (int)(nb > 4096 ? 4096 : nb);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.partial, dst, sp.nslab, ntaps, p.A, p.C, split);
    return nsg_check_launch("wgrad_reduce_kernel");
}
