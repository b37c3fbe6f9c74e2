// Implicit-GEMM convolution forward / data-gradient on the gfx950 matrix cores:
//   fp32 operands -> v_mfma_f32_32x32x2_f32 (exact fp32, the parity mode)
//   bf16 operands -> v_mfma_f32_32x32x16_bf16 (fp32 accumulate, the throughput mode).
// A K-chunk is 128 BYTES of one tap's channels in both modes (32 floats or 64 bf16), so staging, the LDS
// image and the ds_read_b128 pattern are byte-identical; one ds_read_b128 feeds four 32x32x2 MFMAs or one 32x32x16.
//
//   out[pixel(m)][n] = bias[n] + sum over taps t, channels c of  in[pixel(m) shifted by tap t][c] * w[slice(t)][n][c]
//
// One kernel serves nn.Conv2d forward, nn.ConvTranspose2d forward and both of their data
// gradients (reference call sites: src/models.py:150,153,165,168,179,182 and their autograd):
//   MODE 0 ("conv gather"):   input pixel = (ry*stride - pad + kh, rx*stride - pad + kw)
//   MODE 1 ("transposed 4/2/1"): the 4 output parity classes (oy&1, ox&1) each use their own 2x2
//       subset of the 4x4 taps; one class per blockIdx.y, so every row of a tile shares its taps.
//
// Layout / mapping (MI355X-first, not a port of anything):
//   * activations NHWC, so a K-chunk (32 channels of one tap) of a row is 128 contiguous bytes;
//   * block = 256 threads = 4 waves (one per SIMD), tile BM x BN = (WM*TM*32) x (WN*TN*32), each
//     wave owns TM x TN accumulators of 32x32 (v16f each, in AGPRs);
//   * A (gathered input rows) and B (weights, [n][c] so both operands are "row = MFMA row/col,
//     contiguous k") are register-staged global -> LDS with a 36-float row pitch: the
//     ds_read_b128 of 16 different rows then lands on 16 different 4-bank groups (conflict-free);
//   * one lane's ds_read_b128 feeds 4 MFMAs: MFMA q of a chunk-of-8 consumes k = {q, 4+q}
//     (lane half h supplies k = 4h+q) for both operands: a fixed permutation of the reduction
//     order, identical on both sides;
//   * double-buffered LDS; the steady-state loop body is ONE basic block (MODE / RELU are template
//     parameters, predicates are branch-free, the last chunk is peeled) so the compiler can issue
//     the next chunk's address arithmetic, global loads and LDS writes in the shadow of the
//     current chunk's 64-cycle MFMAs; loads are unconditional (clamped address), zero-fill and
//     the fused ReLU are applied when the registers are written to LDS;
//   * epilogue: accumulators -> LDS -> 16-byte row-contiguous stores (the MFMA C layout would
//     otherwise give 4-byte stores, which leaves the store-bound C=1 layers at a third of HBM rate).
#include "nsg_common.h"

namespace {

// (Two variants were measured slower inside the training step and removed: LDS-DMA operand staging -- buffer_load ... lds
// into a ring of three A and two B stages -- and a 256 x 128 tile on eight waves; numbers in DESIGN.md 3.1b.)
template <typename TI, typename TO, int WM, int WN, int TM, int TN, int MODE, bool RELU>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2))) void gather_gemm_kernel(const GatherGemmParams p)
{
    constexpr int LDS_PITCH = 36;               // floats per staged row
    constexpr int EPV = 16 / (int)sizeof(TI);   // elements per 16-byte piece: 4 (fp32) or 8 (bf16)
    constexpr int KC = 8 * EPV;                 // channels per K-chunk: 128 bytes of a row
    constexpr bool BF = sizeof(TI) == 2;
    const TI *__restrict__ gin = reinterpret_cast<const TI *>(p.in);
    const TI *__restrict__ gw = reinterpret_cast<const TI *>(p.w);
    TO *__restrict__ gout = reinterpret_cast<TO *>(p.out);
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int NT = 64 * WM * WN;     // threads: 4 waves
    constexpr int RPP = NT / 8;          // rows staged per pass of the block (8 threads per 128-byte row)
    constexpr int AJ = BM / RPP;         // 16-byte pieces per thread for A
    constexpr int BJ = BN / RPP;
    constexpr int CP = BN + 4;   // epilogue staging pitch (floats)
    static_assert(WM * WN == 4 && BM % RPP == 0 && BN % RPP == 0, "4 waves per block");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // register staging: [2][BM][36] + [2][BN][36]
    constexpr int ASTAGES = 2;
    float *As = smem;                               // [ASTAGES][BM][LDS_PITCH]
    float *Bs = smem + ASTAGES * BM * LDS_PITCH;    // [2][BN][LDS_PITCH]
    // the region serves the operand stages, then the output staging tile (+ the statistics scratch): sized for the larger
    constexpr int STAGE_FLOATS = ASTAGES * BM * LDS_PITCH + 2 * BN * LDS_PITCH;
    constexpr int REGION_FLOATS = STAGE_FLOATS > BM * CP + 512 ? STAGE_FLOATS : BM * CP + 512;
    int *rowoff = reinterpret_cast<int *>(smem + REGION_FLOATS);     // output row offsets [BM]: behind the region
    float *Cs = smem;                         // epilogue: [BM][CP], reuses As/Bs

    const unsigned long long st_entry = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;   // diagnostics
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;

    const int ntiles_n = (p.CO + BN - 1) / BN;
    // Blocks are dealt round-robin over the 8 XCDs (observed, speed only): give each XCD a CONTIGUOUS run of tiles so
    // that the rows a tile shares with its neighbours (the taps' halo) are hits in that XCD's own L2.  Bijective for
    // any grid size (cdna_hip_programming.md, XCD swizzle).
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    const int mtile = bid / ntiles_n;
    const int ntile = bid % ntiles_n;
    const int m0 = mtile * BM;
    const int n0 = ntile * BN;
    const int cls = blockIdx.y;
    const int py = cls >> 1, px = cls & 1;

    // ---- per-thread gather rows: row r_j = tid/8 + 32 j ----
    // Operands are fetched with buffer loads: an offset past the end of the buffer reads as zero, so the conv padding,
    // ragged row tiles and ragged channel chunks cost one select on the OFFSET instead of selects on the data.
    constexpr unsigned ES = sizeof(TI);
    constexpr unsigned OOB = 0xfffffff0u;   // >= any buffer size accepted by the launcher
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<TI *>(gin), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<TI *>(gw), 0, (int)p.w_bytes, 0x00020000);
    const int rsub = tid >> 3;
    const int c4 = (tid & 7) * EPV;          // this thread's 16-byte piece inside a chunk, in elements
    const int ntaps = (MODE == 0) ? p.KH * p.KW : 4;
    unsigned rbase[AJ];    // BYTE offset of (b, iy0, ix0, c4) (wraps for "negative" pixels; only used when the tap is valid)
    unsigned tapmask[AJ];  // bit kh: tap row kh lies inside the image for this row; bit 16+kw: tap column kw does
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int m = m0 + rsub + RPP * j;
        rbase[j] = 0;
        tapmask[j] = 0;
        if (m < p.M) {
            const int b = nsg_div(m, p.div_rhw);            // launch-invariant divisors: multiply-shift, not the ~40-instruction division
            const int rem = m - b * (p.RH * p.RW);
            const int ry = nsg_div(rem, p.div_rw);
            const int rx = rem - ry * p.RW;
            int iy0, ix0;
            if (MODE == 0) { iy0 = ry * p.stride - p.pad; ix0 = rx * p.stride - p.pad_w; }
            else           { iy0 = ry + py;               ix0 = rx + px; }
            rbase[j] = (unsigned)(((b * p.IH + iy0) * p.IW + ix0) * p.CI + c4) * ES;
            const int khl = (MODE == 0) ? p.KH : 2, kwl = (MODE == 0) ? p.KW : 2;
            unsigned tm = 0;
            for (int kh = 0; kh < khl; ++kh) {
                const int iy = (MODE == 0) ? iy0 + kh : iy0 - kh;
                tm |= (unsigned)((iy >= 0) & (iy < p.IH)) << kh;
            }
            for (int kw = 0; kw < kwl; ++kw) {
                const int ix = (MODE == 0) ? ix0 + kw : ix0 - kw;
                tm |= (unsigned)((ix >= 0) & (ix < p.IW)) << (16 + kw);
            }
            tapmask[j] = tm;
        }
    }
    unsigned wbase[BJ];    // BYTE offset of (tap 0, n, c4)
    unsigned wmask[BJ];    // all ones when column n exists, else zero
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int n = n0 + rsub + RPP * j;
        wbase[j] = (unsigned)(n * p.CI + c4) * ES;
        wmask[j] = n < p.CO ? 0xffffffffu : 0u;
    }
    // ---- output row offsets ----
    auto fill_rowoff = [&]() {
    for (int t = tid; t < BM; t += NT) {
        const int m = m0 + t;
        int off = -1;
        if (m < p.M) {
            const int b = nsg_div(m, p.div_rhw);            // launch-invariant divisors: multiply-shift, not the ~40-instruction division
            const int rem = m - b * (p.RH * p.RW);
            const int ry = nsg_div(rem, p.div_rw);
            const int rx = rem - ry * p.RW;
            const int oy = (MODE == 0) ? ry : 2 * ry + py;
            const int ox = (MODE == 0) ? rx : 2 * rx + px;
            if (oy < p.OH && ox < p.OW) off = ((b * p.OH + oy) * p.OW + ox) * p.CO;  // odd extents: last class row/col absent
        }
        rowoff[t] = off;
    }
    };
    fill_rowoff();

    // the epilogue's column group of this thread and its bias values: fetched now, the epilogue must not wait on them
    constexpr int EPO = 16 / (int)sizeof(TO);   // output elements per 16-byte store
    constexpr int NV = BN / EPO;                // 16-byte pieces per tile row; divides the thread count, so a thread keeps its columns
    static_assert(NT % NV == 0, "a thread's column group must not change from row to row");
    const int cq = (tid % NV) * EPO;
    const int col = n0 + cq;
    float bv[EPO];
#pragma unroll
    for (int e = 0; e < EPO; ++e) bv[e] = (p.bias && col + e < p.CO) ? p.bias[col + e] : 0.f;

    const int nchunks = (p.CI + KC - 1) / KC;
    const int nit = ntaps * nchunks;

    v4f ra0[AJ], rb0[BJ], ra1[AJ], rb1[BJ];   // two register stages: loads run two chunks ahead of the MFMAs

    // state of the NEXT chunk to load (advanced incrementally: no divisions in the loop)
    int g_c0 = 0, g_kh = 0, g_kw = 0;   // MODE 0: (kh,kw) of the tap; MODE 1: (a,b2) of the class tap
    auto gload = [&](v4f (&ra)[AJ], v4f (&rb)[BJ]) {
        int dy, dx, ws;
        if (MODE == 0) { dy = g_kh; dx = g_kw; ws = g_kh * p.KW + g_kw; }
        else           { dy = -g_kh; dx = -g_kw; ws = ((1 - py) + 2 * g_kh) * 4 + (1 - px) + 2 * g_kw; }
        const int c0 = g_c0;
        const unsigned tap_b = (unsigned)((dy * p.IW + dx) * p.CI + c0) * ES;   // uniform
        const unsigned wtap_b = (unsigned)(ws * p.CO * p.CI + c0) * ES;         // uniform
        const bool cok = (c0 + c4) < p.CI;
        // both the tap's row bit and its column bit must be set; an all-ones pattern (never a mask value) disables the piece
        const unsigned sel = cok ? ((1u << (g_kh & 15)) | (0x10000u << (g_kw & 15))) : 0xffffffffu;
        const unsigned cm = cok ? 0xffffffffu : 0u;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const unsigned off = ((tapmask[j] & sel) == sel) ? rbase[j] + tap_b : OOB;
            ra[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)off, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const unsigned m = wmask[j] & cm;                       // branch-free: (m ? offset : OOB) as a bit select
            const unsigned off = ((wbase[j] + wtap_b) & m) | (OOB & ~m);
            rb[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)off, 0, 0));
        }
        // advance to the following chunk
        g_c0 += KC;
        const int wrap = g_c0 >= p.CI;
        g_c0 = wrap ? 0 : g_c0;
        g_kw += wrap;
        const int kwlim = (MODE == 0) ? p.KW : 2;
        const int wrap2 = g_kw >= kwlim;
        g_kw = wrap2 ? 0 : g_kw;
        g_kh += wrap2;
    };
    auto lstore = [&](int buf, const v4f (&ra)[AJ], const v4f (&rb)[BJ]) {
        float *a = As + buf * BM * LDS_PITCH;
        float *b = Bs + buf * BN * LDS_PITCH;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            v4f v = ra[j];
            if (RELU) {
                if constexpr (BF) {   // a bf16 is negative exactly when its bit pattern is a negative int16
                    s16x8 sv = __builtin_bit_cast(s16x8, v);
                    const s16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
                    sv = __builtin_elementwise_max(sv, z8);
                    v = __builtin_bit_cast(v4f, sv);
                } else {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
            }
            *reinterpret_cast<v4f *>(a + (rsub + RPP * j) * LDS_PITCH + (tid & 7) * 4) = v;
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<v4f *>(b + (rsub + RPP * j) * LDS_PITCH + (tid & 7) * 4) = rb[j];
    };

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int cur) {
        const float *a_base = As + cur * BM * LDS_PITCH + (wr * TM * 32 + l31) * LDS_PITCH + 4 * h;
        const float *b_base = Bs + cur * BN * LDS_PITCH + (wc * TN * 32 + l31) * LDS_PITCH + 4 * h;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            v4f a[TM], b[TN];
            const int kofs = kk * 8;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const v4f *>(a_base + i * 32 * LDS_PITCH + kofs);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const v4f *>(b_base + j * 32 * LDS_PITCH + kofs);
            if constexpr (BF) {
                // lane (r, h) holds k = 16*kk + 8h + 0..7 of its row: exactly the 32x32x16 operand layout
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j]), __builtin_bit_cast(bf16x8, a[i]),
                                                                           acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][q], a[i][q], acc[i][j], 0, 0, 0);
            }
        }
    };

    unsigned long long st_t0 = 0, st_r0 = 0;
    if (p.stamps) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }

    // Chunk i is computed from LDS buffer i&1 while chunk i+1 waits in registers (stored to the other buffer right
    // after the MFMAs of chunk i are issued) and chunk i+2 is in flight from memory: a 32x32x16 chunk is only ~512
    // matrix-pipe cycles, shorter than a global load, so a single register stage would stall every chunk.
    gload(ra0, rb0);                  // chunk 0
    if (nit > 1) gload(ra1, rb1);     // chunk 1
    lstore(0, ra0, rb0);
    __syncthreads();
    int it = 0;
    for (; it + 2 < nit; it += 2) {   // straight-line body (two chunks): buffer 0 holds chunk it, ra1/rb1 hold chunk it+1
        gload(ra0, rb0);              // chunk it+2
        compute(0);
        lstore(1, ra1, rb1);
        __syncthreads();
        gload(ra1, rb1);              // chunk it+3 (past the end: every offset is out of range, reads zeros, never stored)
        compute(1);
        lstore(0, ra0, rb0);
        __syncthreads();
    }
    compute(0);
    if (it + 1 < nit) {
        lstore(1, ra1, rb1);
        __syncthreads();
        compute(1);
    }
    if (p.stamps && tid == 0) {   // diagnostics: never read by any kernel
        const int sbid = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        p.stamps[4 * sbid] = t1 - st_t0;                                   // main loop, shader cycles
        p.stamps[4 * sbid + 1] = __builtin_amdgcn_s_memrealtime() - st_r0; // main loop, 100 MHz ticks
        p.stamps[4 * sbid + 2] = st_t0 - st_entry;                         // prologue (index arithmetic), shader cycles
        p.stamps[4 * sbid + 3] = t1;                                       // (epilogue = exit stamp - this; written below)
    }
    __syncthreads();   // everyone is done reading As/Bs: the region becomes the output staging tile

    // ---- epilogue.  The MFMAs were issued with the operands swapped (weights as A, pixels as B), so an accumulator tile is
    //      the TRANSPOSED output tile: in the 32x32 C/D layout (column = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
    //      the column is the pixel and the rows are output channels -- each lane holds 4 CONSECUTIVE channels of its pixel per
    //      register group, i.e. 16-byte LDS writes (16 per thread instead of 64 scalar ones).  Same products, same order. ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wr * TM * 32 + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = wc * TN * 32 + j * 32 + 8 * g + 4 * h;
                *reinterpret_cast<v4f *>(Cs + row * CP + col) = v4f{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            }
    }
    __syncthreads();
    if (p.stamps && tid == 0) {   // diagnostics: cycles from the last MFMA issue to the output tile standing in LDS
        const int sbid = blockIdx.y * gridDim.x + blockIdx.x;
        p.stamps[4 * sbid + 1] = __builtin_amdgcn_s_memtime() - p.stamps[4 * sbid + 3];
    }
    // ---- optional: BatchNorm batch statistics of this output tile, merged across tiles by
    //      bn_stats_from_tiles -- saves a full read pass over the conv output.  One LDS pass: sums of
    //      (v - pivot) and (v - pivot)^2 with pivot = the tile's first row (a sample of the column, so
    //      no cancellation); written as (count, sum v, sum (v - mean_tile)^2).  Done BEFORE the output stores: its barrier
    //      would otherwise wait for every store of the tile to be acknowledged (vmcnt(0)) ----
    if constexpr (NT == 256) if (p.stats != nullptr) {
        constexpr int PARTS = 256 / BN;          // threads per column
        constexpr int RPP = BM / PARTS;          // rows per part
        float *red = smem + BM * CP;             // [2][256] scratch behind the staging tile
        const int colw = tid % BN, part = tid / BN;
        const bool all_valid = (MODE == 0) ? (m0 + BM <= p.M) : false;   // conv-mode interior tile: every row is real
        const float pivot = Cs[colw];            // row 0 of the tile (always a real row when the tile has any)
        float s1 = 0.f, s2 = 0.f, cnt = 0.f;
        if (all_valid) {
#pragma unroll 8
            for (int r = part * RPP; r < (part + 1) * RPP; ++r) {
                const float dlt = Cs[r * CP + colw] - pivot;
                s1 += dlt;
                s2 += dlt * dlt;
            }
            cnt = (float)RPP;
        } else {
            for (int r = part * RPP; r < (part + 1) * RPP; ++r)
                if (rowoff[r] >= 0) {
                    const float dlt = Cs[r * CP + colw] - pivot;
                    s1 += dlt;
                    s2 += dlt * dlt;
                    cnt += 1.f;
                }
        }
        red[tid] = s1;
        red[256 + tid] = s2;
        __syncthreads();
        const int colg = n0 + colw;
        if (part == 0 && colg < p.CO) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) { t1 += red[q * BN + colw]; t2 += red[256 + q * BN + colw]; }
            float tcnt = cnt;
            if (!all_valid) {   // counts differ per part only in edge tiles: recount exactly
                tcnt = 0.f;
                for (int r = 0; r < BM; ++r) tcnt += rowoff[r] >= 0 ? 1.f : 0.f;
            } else {
                tcnt = (float)BM;
            }
            const int ntm = gridDim.x / ntiles_n;
            float *dst = p.stats + (size_t)(cls * ntm + mtile) * 3 * p.CO;
            const float bv = p.bias ? p.bias[colg] : 0.f;
            const float inv = tcnt > 0.f ? 1.f / tcnt : 0.f;
            dst[colg] = tcnt;
            dst[p.CO + colg] = tcnt * (pivot + bv) + t1;            // sum of the (biased) outputs over the tile's rows
            dst[2 * p.CO + colg] = fmaxf(t2 - t1 * t1 * inv, 0.f);  // sum of squares about the tile mean
        }
    }
    const bool tanh_out = (p.flags & NSG_TANH_OUT) != 0;
    const bool relu_out = (p.flags & NSG_RELU_OUT) != 0;
    const bool vec_store = ((p.CO % EPO) == 0) && nsg_aligned16_dev(p.out);
    const bool full_vec = vec_store && col + EPO - 1 < p.CO;
    const TO *__restrict__ epi_add = reinterpret_cast<const TO *>(p.epi_add);     // (launcher: only with full 16-byte pieces)
    const TO *__restrict__ epi_mask = reinterpret_cast<const TO *>(p.epi_mask);
    if (col < p.CO) {
        constexpr int RSTEP = NT / NV;           // tile rows covered per pass of the block
        constexpr int RI = BM / RSTEP;           // passes
        constexpr int GRP = RI < 8 ? RI : 8;     // the fused add / mask operands of GRP passes are fetched together
        const bool fuse_add = full_vec && epi_add != nullptr, fuse_mask = full_vec && epi_mask != nullptr;
        for (int g0 = 0; g0 < RI; g0 += GRP) {
            v4f addv[GRP], maskv[GRP];
            if (fuse_add) {
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    const int off = rowoff[tid / NV + (g0 + i) * RSTEP];
                    addv[i] = *reinterpret_cast<const v4f *>(epi_add + (size_t)(off < 0 ? 0 : off) + col);   // clamped, unconditional
                }
            }
            if (fuse_mask) {
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    const int off = rowoff[tid / NV + (g0 + i) * RSTEP];
                    maskv[i] = *reinterpret_cast<const v4f *>(epi_mask + (size_t)(off < 0 ? 0 : off) + col);
                }
            }
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int row = tid / NV + (g0 + i) * RSTEP;
                const int off = rowoff[row];
                if (off < 0) continue;
                float v[EPO];
#pragma unroll
                for (int e = 0; e < EPO; e += 4) {
                    const v4f t = *reinterpret_cast<const v4f *>(Cs + row * CP + cq + e);
                    v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
                }
#pragma unroll
                for (int e = 0; e < EPO; ++e) v[e] += bv[e];
                if (tanh_out) {
#pragma unroll
                    for (int e = 0; e < EPO; ++e) v[e] = tanhf(v[e]);
                }
#pragma unroll
                for (int e = 0; e < EPO; ++e) v[e] = relu_out ? fmaxf(v[e], 0.f) : v[e];   // (a select keeps NaNs visible)
                if (fuse_add) {      // + the skip-path gradient (the ResBlock's residual add, backward)
                    float t[EPO];
                    Elem<TO>::unpack16(addv[i], t);
#pragma unroll
                    for (int e = 0; e < EPO; ++e) v[e] += t[e];
                }
                if (fuse_mask) {     // gradient through the ReLU whose OUTPUT is epi_mask (x > 0 <=> relu(x) > 0)
                    float t[EPO];
                    Elem<TO>::unpack16(maskv[i], t);
#pragma unroll
                    for (int e = 0; e < EPO; ++e) v[e] = t[e] > 0.f ? v[e] : 0.f;
                }
                TO *dst = gout + (size_t)off + col;
                if (full_vec) {
                    Elem<TO>::store16(dst, v);
                } else {
#pragma unroll
                    for (int e = 0; e < EPO; ++e)
                        if (col + e < p.CO) Elem<TO>::put(dst + e, v[e]);
                }
            }
        }
    }

    if (p.stamps) {   // diagnostics: epilogue cycles of this workgroup (all its stores issued)
        __syncthreads();
        if (tid == 0) {
            const int sbid = blockIdx.y * gridDim.x + blockIdx.x;
            p.stamps[4 * sbid + 3] = __builtin_amdgcn_s_memtime() - p.stamps[4 * sbid + 3];
        }
    }
}

template <typename TI, typename TO, int WM, int WN, int TM, int TN, int MODE, bool RELU>
int launch_one(const GatherGemmParams &p, hipStream_t s)
{
    constexpr int LDS_PITCH = 36;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t STAGE_FLOATS = (size_t)2 * BM * LDS_PITCH + (size_t)2 * BN * LDS_PITCH;
    constexpr size_t CS_FLOATS = (size_t)BM * (BN + 4) + 512;
    const size_t lds = (STAGE_FLOATS > CS_FLOATS ? STAGE_FLOATS : CS_FLOATS) * sizeof(float) + BM * sizeof(int);
    const int ntn = (p.CO + BN - 1) / BN;
    const int64_t ntm = nsg_cdiv(p.M, BM);
    const int64_t gx = ntm * ntn;
    if (gx <= 0 || gx > 0x7fffffff) return nsg_fail(NSG_E_UNSUPPORTED, "gather_gemm: grid too large");
    dim3 grid((unsigned)gx, MODE == 0 ? 1 : 4, 1);
    static LdsOptIn once;   // > 64 KiB of dynamic LDS must be opted into once per kernel and device
    if (lds > 65536) {
        const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&gather_gemm_kernel<TI, TO, WM, WN, TM, TN, MODE, RELU>)}, lds, "gather_gemm");
        if (rc != NSG_OK) return rc;
    }
    hipLaunchKernelGGL((gather_gemm_kernel<TI, TO, WM, WN, TM, TN, MODE, RELU>), grid, dim3(64 * WM * WN), lds, s, p);
    return nsg_check_launch("gather_gemm");
}

template <typename TI, typename TO, int WM, int WN, int TM, int TN>
int launch_cfg(const GatherGemmParams &p, hipStream_t s)
{
    const bool relu = (p.flags & NSG_RELU_IN) != 0;
    if (p.mode == 0) return relu ? launch_one<TI, TO, WM, WN, TM, TN, 0, true>(p, s) : launch_one<TI, TO, WM, WN, TM, TN, 0, false>(p, s);
    return relu ? launch_one<TI, TO, WM, WN, TM, TN, 1, true>(p, s) : launch_one<TI, TO, WM, WN, TM, TN, 1, false>(p, s);
}

template <typename TI, typename TO>
int launch_typed(const GatherGemmParams &p, hipStream_t s)
{
    if (p.CO > 64) return launch_cfg<TI, TO, 2, 2, 2, 2>(p, s);   // 128 x 128
    if (p.CO > 32) return launch_cfg<TI, TO, 2, 2, 2, 1>(p, s);   // 128 x 64
    return launch_cfg<TI, TO, 4, 1, 1, 1>(p, s);                  // 128 x 32
}

}  // namespace

int nsg_gather_gemm_row_tiles(const GatherGemmParams &p)
{
    GatherGemmParams q = p;         // (the byte sizes only matter to the launch itself)
    if (!q.stats) q.stats = reinterpret_cast<float *>(16);
    const int rec = nsg_patch_gemm_stat_records(q);
    if (rec > 0) return rec;        // gemm_patch.hip: one record per workgroup
    return (int)nsg_cdiv(p.M, 128) * (p.mode == 0 ? 1 : 4);
}

int nsg_launch_gather_gemm(const GatherGemmParams &p_in, hipStream_t s)
{
    GatherGemmParams p = p_in;
    const uint64_t rows = (uint64_t)p.B * p.RH * p.RW;       // callers form p.M in 32 bits
    if (rows > 0x7fffffffull || (uint64_t)(int64_t)p.M != rows)
        return nsg_fail(NSG_E_UNSUPPORTED, "gather_gemm: %llu output rows overflow 31 bits (split the batch)", (unsigned long long)rows);
    if (p.M <= 0) return NSG_OK;
    const int epv = p.in_dtype == NSG_BF16 ? 8 : 4;
    const uint64_t es = p.in_dtype == NSG_BF16 ? 2 : 4;
    const uint64_t in_bytes = (uint64_t)p.B * p.IH * p.IW * p.CI * es;
    const uint64_t w_bytes = (uint64_t)(p.mode == 0 ? p.KH * p.KW : 16) * p.CO * p.CI * es;
    if (in_bytes >= 0xfffffff0ull || w_bytes >= 0xfffffff0ull)   // 32-bit byte offsets; 0xfffffff0 is the kernel's "reads as zero" offset
        return nsg_fail(NSG_E_UNSUPPORTED, "gather_gemm: operand of %llu bytes (limit 4 GiB per tensor: split the batch)", (unsigned long long)in_bytes);
    // output rows are addressed by 32-bit ELEMENT offsets (rowoff; the fused add / ReLU-mask operands share them): a transposed
    // conv's output is 4x its input, so it can pass the byte limits above and still not fit (B ~ 820 at D = 128, 80 x 1024)
    const uint64_t out_elems = (uint64_t)p.B * p.OH * p.OW * p.CO;
    if (out_elems >= 0x80000000ull)
        return nsg_fail(NSG_E_UNSUPPORTED, "gather_gemm: output of %llu elements (limit 2^31 per tensor: split the batch)", (unsigned long long)out_elems);
    if (p.KH > 15 || p.KW > 15) return nsg_fail(NSG_E_UNSUPPORTED, "gather_gemm: kernel extent above 15");
    if (p.epi_add || p.epi_mask) {
        const int epo = p.out_dtype == NSG_BF16 ? 8 : 4;
        if (p.CO % epo != 0 || !nsg_aligned16(p.out) || (p.epi_add && !nsg_aligned16(p.epi_add)) || (p.epi_mask && !nsg_aligned16(p.epi_mask)))
            return nsg_fail(NSG_E_UNSUPPORTED, "gather_gemm: the fused add / ReLU-mask epilogue needs C_out %% %d == 0 and 16-byte aligned tensors", epo);
    }
    p.in_bytes = (unsigned)in_bytes;
    p.w_bytes = (unsigned)w_bytes;
    p.div_rw = nsg_fastdiv((uint32_t)p.RW);
    p.div_rhw = nsg_fastdiv((uint32_t)p.RH * (uint32_t)p.RW);
    if (p.CI % epv != 0) return nsg_fail(NSG_E_UNSUPPORTED, "gather_gemm: C_in=%d not a multiple of %d", p.CI, epv);
    if (!nsg_aligned16(p.in) || !nsg_aligned16(p.w)) return nsg_fail(NSG_E_INVALID, "gather_gemm: operands must be 16-byte aligned");
    {
        bool handled = false;
        const int rc = nsg_launch_patch_gemm(p, s, &handled);
        if (handled || rc != NSG_OK) return rc;
    }
    if (p.in_dtype == NSG_F32 && p.out_dtype == NSG_F32) return launch_typed<float, float>(p, s);
    if (p.in_dtype == NSG_BF16 && p.out_dtype == NSG_BF16) return launch_typed<bf16_t, bf16_t>(p, s);
    if (p.in_dtype == NSG_BF16 && p.out_dtype == NSG_F32) return launch_typed<bf16_t, float>(p, s);
    return nsg_fail(NSG_E_UNSUPPORTED, "gather_gemm: fp32 operands with bf16 output are not supported");
}
