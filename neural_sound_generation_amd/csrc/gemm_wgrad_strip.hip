// Weight gradient of the bf16 3x3/1 and 4x4/2 (plain and transposed) convolutions, one KERNEL ROW per workgroup
// ("row-strip multi-tap"; autograd of src/models.py:150,168,179):
//
//   G[kh][kw][a][c] = sum over pixels (b, y, x) of the conv-OUTPUT grid   P[b][y][x][a] * Q[b][y*s - pad + kh][x*s - pad + kw][c]
//
// gemm_wgrad.hip's bf16 kernel gives one TAP to a workgroup, so a chunk of 64 pixels costs 2 x 16 KB of VGPR -> LDS staging
// per 64 MFMAs and P (identical for all taps) is staged taps-many times: it ran LDS-store-bound at 0.30 of the bf16 peak.
// Here a workgroup owns a kernel row kh and walks strips of 64 CONSECUTIVE pixels of one image row: the P strip is staged
// once for the row's KW taps, and the Q strip (64*s - s + KW pixels of input row y*s - pad + kh) serves them all -- tap kw is
// the same LDS rows shifted by kw (stride 2: the strip is stored de-interleaved by pixel parity, so tap kw reads plane kw & 1
// shifted by kw >> 1).  Staging per MFMA drops 2.6-3x; the rest is gemm_wgrad.hip's recipe: register-staged global -> LDS with
// two register stages (loads two chunks ahead), plain [pixel][channel] LDS rows with a 2*T + 64 byte pitch read with the
// transposing ds_read_b64_tr_b16 (the reduction index -- the pixel -- is the ROW of both NHWC operands), fp32 accumulation,
// per-slab partial tiles merged in fixed order by wgrad_reduce_kernel (bitwise reproducible).
//
// Block = 8 waves (2 per SIMD, one workgroup per CU): wave (wr, wc) owns channels a = 64 wr .. + 63 (two 32-blocks) x
// c = 32 wc .. + 31 for every tap of the row: 2 x KW accumulators of 32 x 32.
#include "nsg_common.h"

namespace {

NSG_DIAG_SWITCH(int, g_wgrad_strip, 3)    // nsg_debug_set_wgrad_strip (diagnostics library only): bit 0 = the bf16 form, bit 1 = the fp32 form; 0 sends everything back to gemm_wgrad.hip's per-tap kernels (A/B runs)

constexpr int SP = 64;              // pixels per strip chunk (4 MFMA k-steps of 16)
constexpr int PITCH = 128 + 32;     // LDS row pitch in bf16 elements: 2 * 128 + 64 bytes (conflict-free transposing reads)

__device__ __forceinline__ s16x4 strip_tr_read(const bf16_t *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
}

struct StripParams {
    const bf16_t *P;        // [B][PH][PW][A]
    const bf16_t *Q;        // [B][QH][QW][C]
    float *partial;         // [nslab][KH*KW][A][C]
    int B, PH, PW, A, QH, QW, C;
    int KH, pad, pad_w;
    int cpr;                // strip chunks per image row = ceil(PW / 64)
    int nchunks;            // B * PH * cpr
    int chunks_per_slab, nslab;
    FastDiv div_cpr, div_rows;      // chunk -> (row, chunk in row); row -> (b, y)
    unsigned p_bytes, q_bytes;
};

// KW taps per kernel row, S = stride (1 or 2)
template <int KW, int S>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void wgrad_strip_bf16(const StripParams p)
{
    constexpr int QS = (SP - 1) * S + KW;               // pixels of the Q strip: 66 (3x3/1) or 130 (4x4/2)
    constexpr int QROWS = S == 1 ? QS : 2 * ((QS + 1) / 2);   // LDS rows of a Q stage (stride 2: two parity planes of (QS + 1) / 2 rows)
    constexpr int QPLANE = (QS + 1) / 2;
    constexpr int PPIECES = SP * 16, QPIECES = QS * 16;  // 16-byte pieces per stage (128 channels = 16 pieces per pixel)
    constexpr int PJ = PPIECES / 512;                    // = 2
    constexpr int QJ = (QPIECES + 511) / 512;            // 3 (66 pixels) or 5 (130 pixels); the last one is mostly idle
    constexpr unsigned OOB = 0xfffffff0u;

    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.P), 0, (int)p.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.Q), 0, (int)p.q_bytes, 0x00020000);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    bf16_t *Ps = reinterpret_cast<bf16_t *>(smem);       // [2][SP][PITCH]
    bf16_t *Qs = Ps + 2 * SP * PITCH;                    // [2][QROWS][PITCH]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int h = lane >> 5;
    // transposing read: lane = 32 h + 16 g16 + 4 q + pp supplies the address of row q, columns 4 pp .. + 3 of its group's block
    const int g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;

    // The KH workgroups of one slab read the same P rows and (shifted) Q rows.  Workgroups are dealt round-robin over the 8
    // XCDs (observed, speed only), so ids 8 apart share an L2: id = ((slab / 8) * KH + kh) * 8 + slab % 8 puts a slab's
    // kernel rows on one XCD, started together -- each row of the two tensors then leaves HBM once, not KH times (PMC:
    // 1.04 GB per launch of a 3x3 before, against 0.34 GB algorithmic).
    const int lin = blockIdx.x;
    const int kh = (lin >> 3) % p.KH;
    const int slab = ((lin >> 3) / p.KH) * 8 + (lin & 7);
    if (slab >= p.nslab) return;
    const int ctiles = p.C >> 7;
    const int a0 = (blockIdx.z / ctiles) * 128;
    const int c0 = (blockIdx.z % ctiles) * 128;

    const int ch_beg = slab * p.chunks_per_slab;
    const int ch_end = min(p.nchunks, ch_beg + p.chunks_per_slab);
    const int nchunk = ch_end - ch_beg;

    // this thread's pieces: piece f = tid + 512 j -> pixel f >> 4, channels 8 (f & 15) .. + 7
    const int pc8 = (tid & 15) * 8;
    const int px0 = tid >> 4;                             // + 32 j

    v4f rp0[PJ], rq0[QJ], rp1[PJ], rq1[QJ];              // two register stages (loads run two chunks ahead of the MFMAs)
    int g_ch = ch_beg;                                    // next chunk to load

    auto gload = [&](v4f (&rp)[PJ], v4f (&rq)[QJ]) {
        const int ch = g_ch;
        const bool live = ch < ch_end;                    // past the slab: every offset is out of range (reads zeros, never stored)
        const int row = nsg_div(live ? ch : 0, p.div_cpr);
        const int x0 = ((live ? ch : 0) - row * p.cpr) * SP;
        const int b = nsg_div(row, p.div_rows);
        const int y = row - b * p.PH;
        const int qy = y * S - p.pad + kh;
        const bool qrow_ok = live & (qy >= 0) & (qy < p.QH);
        const unsigned pbase = (unsigned)((b * p.PH + y) * p.PW + x0) * (unsigned)p.A * 2u + (unsigned)(a0 + pc8) * 2u;
        const int qx0 = x0 * S - p.pad_w;
        const unsigned qrow = (unsigned)((b * p.QH + qy) * p.QW) * (unsigned)p.C * 2u + (unsigned)(c0 + pc8) * 2u;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int px = px0 + 32 * j;
            const bool ok = live & (x0 + px < p.PW);
            rp[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_p, (int)(ok ? pbase + (unsigned)(px * p.A) * 2u : OOB), 0, 0));
        }
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const int jj = px0 + 32 * j;                  // pixel of the strip
            const int qx = qx0 + jj;
            const bool ok = qrow_ok & (jj < QS) & (qx >= 0) & (qx < p.QW);
            rq[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_q, (int)(ok ? qrow + (unsigned)(qx * p.C) * 2u : OOB), 0, 0));
        }
        g_ch += 1;
    };
    auto lstore = [&](int buf, const v4f (&rp)[PJ], const v4f (&rq)[QJ]) {
        bf16_t *ps = Ps + buf * SP * PITCH;
        bf16_t *qs = Qs + buf * QROWS * PITCH;
#pragma unroll
        for (int j = 0; j < PJ; ++j) *reinterpret_cast<v4f *>(ps + (px0 + 32 * j) * PITCH + pc8) = rp[j];
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const int jj = px0 + 32 * j;
            const int lrow = S == 1 ? jj : (jj & 1) * QPLANE + (jj >> 1);
            if (QPIECES % 512 == 0 || j + 1 < QJ || jj < QS) *reinterpret_cast<v4f *>(qs + lrow * PITCH + pc8) = rq[j];
        }
    };

    v16f acc[2][KW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < KW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    auto compute = [&](int cur) {
        // this lane's address inside a (k-step, 32-channel block): row = 8 h + q (+ 4 for the second half), column = 16 g16 + 4 pp
        const bf16_t *pbase = Ps + cur * SP * PITCH + (8 * h + q) * PITCH + wr * 64 + 16 * g16 + 4 * pp;
        const bf16_t *qbase = Qs + cur * QROWS * PITCH + (8 * h + q) * PITCH + wc * 32 + 16 * g16 + 4 * pp;
#pragma unroll
        for (int ks = 0; ks < SP / 16; ++ks) {
            s16x8 a[2], b[KW];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const s16x4 lo = strip_tr_read(pbase + (16 * ks) * PITCH + i * 32);
                const s16x4 hi = strip_tr_read(pbase + (16 * ks + 4) * PITCH + i * 32);
                a[i] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int t = 0; t < KW; ++t) {
                // pixel i of the strip meets Q pixel i * S + t: stride 1 -> row i + t; stride 2 -> plane t & 1, row i + (t >> 1)
                const int roff = S == 1 ? t : (t & 1) * QPLANE + (t >> 1);
                const s16x4 lo = strip_tr_read(qbase + (16 * ks + roff) * PITCH);
                const s16x4 hi = strip_tr_read(qbase + (16 * ks + roff + 4) * PITCH);
                b[t] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < KW; ++t)
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[t]),
                                                                       acc[i][t], 0, 0, 0);
        }
    };

    // chunk i computed from LDS buffer i & 1, chunk i + 1 waiting in registers, chunk i + 2 in flight
    if (nchunk > 0) {
        gload(rp0, rq0);
        if (nchunk > 1) gload(rp1, rq1);
        lstore(0, rp0, rq0);
        __syncthreads();
        int ch = 0;
        for (; ch + 2 < nchunk; ch += 2) {
            gload(rp0, rq0);          // chunk ch + 2
            compute(0);
            lstore(1, rp1, rq1);
            __syncthreads();
            gload(rp1, rq1);          // chunk ch + 3
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

    const int ntaps = p.KH * KW;
    const int l31 = lane & 31;
    const int c = c0 + wc * 32 + l31;
#pragma unroll
    for (int t = 0; t < KW; ++t) {
        float *dst = p.partial + ((size_t)(slab * ntaps + kh * KW + t) * p.A) * p.C;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int a = a0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                dst[(size_t)a * p.C + c] = acc[i][t][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// The fp32 parity mode's form: the same strips on v_mfma_f32_32x32x2_f32 (exact fp32 products, one rounding per product).
// An MFMA operand is ONE float per lane -- lane (i = lane & 31, k = lane >> 5) reads element [pixel k][channel i] of a plain
// [pixel][128 channels] LDS row (32 consecutive floats: conflict-free) -- so no transposing reads and no parity planes:
// tap kw of strip pixel i is row i * S + kw.  SPF pixels per chunk: 64 (3x3/1: 2 x (32 + 33) KB of LDS) or 32 (4x4/2).
// ------------------------------------------------------------------------------------------------
template <int KW, int S>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void wgrad_strip_f32(const StripParams p)
{
    constexpr int SPF = S == 1 ? 64 : 32;
    constexpr int QS = (SPF - 1) * S + KW;               // 66 pixels either way
    constexpr int ROWF = 128;                            // floats per LDS row (no padding needed: a read is one row's 32 consecutive floats)
    constexpr int PPIECES = SPF * 32, QPIECES = QS * 32; // 16-byte pieces per stage (128 fp32 channels = 32 pieces per pixel)
    constexpr int PJ = PPIECES / 512;                    // 4 or 2
    constexpr int QJ = (QPIECES + 511) / 512;            // 5 (the last one is mostly idle)
    constexpr unsigned OOB = 0xfffffff0u;

    const float *gP = reinterpret_cast<const float *>(p.P);
    const float *gQ = reinterpret_cast<const float *>(p.Q);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gP), 0, (int)p.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gQ), 0, (int)p.q_bytes, 0x00020000);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ps = smem;                                    // [2][SPF][ROWF]
    float *Qs = smem + 2 * SPF * ROWF;                   // [2][QS][ROWF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, hk = lane >> 5;

    const int lin = blockIdx.x;                          // a slab's kernel rows share an XCD (see wgrad_strip_bf16)
    const int kh = (lin >> 3) % p.KH;
    const int slab = ((lin >> 3) / p.KH) * 8 + (lin & 7);
    if (slab >= p.nslab) return;
    const int ctiles = p.C >> 7;
    const int a0 = (blockIdx.z / ctiles) * 128;
    const int c0 = (blockIdx.z % ctiles) * 128;

    const int ch_beg = slab * p.chunks_per_slab;
    const int ch_end = min(p.nchunks, ch_beg + p.chunks_per_slab);
    const int nchunk = ch_end - ch_beg;

    // this thread's pieces: piece f = tid + 512 j -> pixel f >> 5, channels 4 (f & 31) .. + 3
    const int pc4 = (tid & 31) * 4;
    const int px0 = tid >> 5;                             // + 16 j

    v4f rp0[PJ], rq0[QJ], rp1[PJ], rq1[QJ];
    int g_ch = ch_beg;
    auto gload = [&](v4f (&rp)[PJ], v4f (&rq)[QJ]) {
        const int ch = g_ch;
        const bool live = ch < ch_end;
        const int row = nsg_div(live ? ch : 0, p.div_cpr);
        const int x0 = ((live ? ch : 0) - row * p.cpr) * SPF;
        const int b = nsg_div(row, p.div_rows);
        const int y = row - b * p.PH;
        const int qy = y * S - p.pad + kh;
        const bool qrow_ok = live & (qy >= 0) & (qy < p.QH);
        const unsigned pbase = (unsigned)((b * p.PH + y) * p.PW + x0) * (unsigned)p.A * 4u + (unsigned)(a0 + pc4) * 4u;
        const int qx0 = x0 * S - p.pad_w;
        const unsigned qrow = (unsigned)((b * p.QH + qy) * p.QW) * (unsigned)p.C * 4u + (unsigned)(c0 + pc4) * 4u;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int px = px0 + 16 * j;
            const bool ok = live & (x0 + px < p.PW);
            rp[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_p, (int)(ok ? pbase + (unsigned)(px * p.A) * 4u : OOB), 0, 0));
        }
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const int jj = px0 + 16 * j;
            const int qx = qx0 + jj;
            const bool ok = qrow_ok & (jj < QS) & (qx >= 0) & (qx < p.QW);
            rq[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_q, (int)(ok ? qrow + (unsigned)(qx * p.C) * 4u : OOB), 0, 0));
        }
        g_ch += 1;
    };
    auto lstore = [&](int buf, const v4f (&rp)[PJ], const v4f (&rq)[QJ]) {
        float *ps = Ps + buf * SPF * ROWF;
        float *qs = Qs + buf * QS * ROWF;
#pragma unroll
        for (int j = 0; j < PJ; ++j) *reinterpret_cast<v4f *>(ps + (px0 + 16 * j) * ROWF + pc4) = rp[j];
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const int jj = px0 + 16 * j;
            if (QPIECES % 512 == 0 || j + 1 < QJ || jj < QS) *reinterpret_cast<v4f *>(qs + jj * ROWF + pc4) = rq[j];
        }
    };

    v16f acc[2][KW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < KW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    auto compute = [&](int cur) {
        const float *pbase = Ps + cur * SPF * ROWF + hk * ROWF + wr * 64 + l31;
        const float *qbase = Qs + cur * QS * ROWF + hk * S * ROWF + wc * 32 + l31;
#pragma unroll 4
        for (int ks = 0; ks < SPF / 2; ++ks) {           // two strip pixels per MFMA step: lanes 0-31 pixel 2 ks, lanes 32-63 pixel 2 ks + 1
            float a[2], b[KW];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = pbase[(2 * ks) * ROWF + 32 * i];
#pragma unroll
            for (int t = 0; t < KW; ++t) b[t] = qbase[((2 * ks) * S + t) * ROWF];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < KW; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[t], acc[i][t], 0, 0, 0);
        }
    };

    if (nchunk > 0) {
        gload(rp0, rq0);
        if (nchunk > 1) gload(rp1, rq1);
        lstore(0, rp0, rq0);
        __syncthreads();
        int ch = 0;
        for (; ch + 2 < nchunk; ch += 2) {
            gload(rp0, rq0);
            compute(0);
            lstore(1, rp1, rq1);
            __syncthreads();
            gload(rp1, rq1);
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

    const int ntaps = p.KH * KW;
    const int c = c0 + wc * 32 + l31;
#pragma unroll
    for (int t = 0; t < KW; ++t) {
        float *dst = p.partial + ((size_t)(slab * ntaps + kh * KW + t) * p.A) * p.C;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int a = a0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hk;
                dst[(size_t)a * p.C + c] = acc[i][t][r];
            }
    }
}

template <int KW, int S>
int launch_strip_f32(const StripParams &p, int nslab, hipStream_t s)
{
    constexpr int SPF = S == 1 ? 64 : 32;
    constexpr int QS = (SPF - 1) * S + KW;
    const size_t lds = (size_t)2 * (SPF + QS) * 128 * sizeof(float);
    static LdsOptIn once;
    if (lds > 65536) {
        const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&wgrad_strip_f32<KW, S>)}, lds, "wgrad_strip_f32");
        if (rc != NSG_OK) return rc;
    }
    dim3 grid((unsigned)(((nslab + 7) / 8) * 8 * p.KH), 1, (unsigned)((p.A >> 7) * (p.C >> 7)));
    hipLaunchKernelGGL((wgrad_strip_f32<KW, S>), grid, dim3(512), lds, s, p);
    return nsg_check_launch("wgrad_strip_f32");
}

template <int KW, int S>
int launch_strip(const StripParams &p, int nslab, hipStream_t s)
{
    constexpr int QS = (SP - 1) * S + KW;
    constexpr int QROWS = S == 1 ? QS : 2 * ((QS + 1) / 2);
    const size_t lds = (size_t)2 * (SP + QROWS) * PITCH * sizeof(bf16_t);
    static LdsOptIn once;
    if (lds > 65536) {
        const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&wgrad_strip_bf16<KW, S>)}, lds, "wgrad_strip");
        if (rc != NSG_OK) return rc;
    }
    dim3 grid((unsigned)(((nslab + 7) / 8) * 8 * p.KH), 1, (unsigned)((p.A >> 7) * (p.C >> 7)));
    hipLaunchKernelGGL((wgrad_strip_bf16<KW, S>), grid, dim3(512), lds, s, p);
    return nsg_check_launch("wgrad_strip_bf16");
}

}  // namespace

#ifdef NSG_DIAG
extern "C" NSG_API void nsg_debug_set_wgrad_strip(int on) { g_wgrad_strip = on; }
#endif

// slabs of the strip kernel for a layer with `ntaps` taps: one resident round of the 256 CUs (one workgroup each)
int nsg_wgrad_strip_slabs(int ntaps, int A, int C)
{
    const int kh = ntaps == 16 ? 4 : 3;
    const int tiles = (A >> 7) * (C >> 7);
    int n = 256 / (kh * (tiles > 0 ? tiles : 1));
    n &= ~7;                                             // whole groups of 8 slabs (one per XCD)
    return n < 8 ? 8 : n;
}

bool nsg_wgrad_strip_applicable(const WgradParams &p)
{
    if (!g_wgrad_strip || p.onehot || p.q_mean || p.relu_p || p.relu_q) return false;
    if (p.dtype == NSG_BF16 ? !(g_wgrad_strip & 1) : !(p.dtype == NSG_F32 && (g_wgrad_strip & 2))) return false;   // bit 0: bf16 form, bit 1: fp32 form
    if (p.A % 128 != 0 || p.C % 128 != 0) return false;
    const bool k33 = p.KH == 3 && p.KW == 3 && p.stride == 1;
    const bool k44 = p.KH == 4 && p.KW == 4 && p.stride == 2;
    return k33 || k44;
}

// Runs the launch (partial slabs only; the caller reduces them); *nslab = slabs written.
int nsg_launch_wgrad_strip(const WgradParams &w, int *nslab_out, hipStream_t s)
{
    StripParams p = {};
    const bool f32 = w.dtype == NSG_F32;
    const int sp = f32 && w.stride == 2 ? 32 : SP;       // pixels per strip chunk
    p.P = reinterpret_cast<const bf16_t *>(w.P);
    p.Q = reinterpret_cast<const bf16_t *>(w.Q);
    p.partial = w.partial;
    p.B = w.B; p.PH = w.PH; p.PW = w.PW; p.A = w.A; p.QH = w.QH; p.QW = w.QW; p.C = w.C;
    p.KH = w.KH; p.pad = w.pad; p.pad_w = w.pad_w;
    p.cpr = (int)nsg_cdiv(w.PW, sp);
    const int64_t nch = (int64_t)w.B * w.PH * p.cpr;
    if (nch > 0x3fffffff) return nsg_fail(NSG_E_UNSUPPORTED, "wgrad_strip: too many strips");
    p.nchunks = (int)nch;
    int nslab = nsg_wgrad_strip_slabs(w.KH * w.KW, w.A, w.C);
    if (nslab > p.nchunks) nslab = p.nchunks;
    p.chunks_per_slab = (int)nsg_cdiv(p.nchunks, nslab);
    nslab = (int)nsg_cdiv(p.nchunks, p.chunks_per_slab);
    p.div_cpr = nsg_fastdiv((uint32_t)p.cpr);
    p.div_rows = nsg_fastdiv((uint32_t)w.PH);
    p.p_bytes = w.p_bytes; p.q_bytes = w.q_bytes;
    if (f32) {
        const uint64_t pb = (uint64_t)w.Mp * w.A * 4, qb = (uint64_t)w.B * w.QH * w.QW * w.C * 4;
        if (pb >= 0xfffffff0ull || qb >= 0xfffffff0ull) return nsg_fail(NSG_E_UNSUPPORTED, "wgrad_strip: operand larger than 4 GiB: split the batch");
        p.p_bytes = (unsigned)pb; p.q_bytes = (unsigned)qb;
    }
    p.nslab = nslab;
    *nslab_out = nslab;
    if (f32) return w.KW == 3 ? launch_strip_f32<3, 1>(p, nslab, s) : launch_strip_f32<4, 2>(p, nslab, s);
    if (w.KW == 3) return launch_strip<3, 1>(p, nslab, s);
    return launch_strip<4, 2>(p, nslab, s);
}
