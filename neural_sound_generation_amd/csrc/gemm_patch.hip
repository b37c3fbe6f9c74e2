// Implicit-GEMM convolution on the bf16 matrix cores with the INPUT PATCH staged once in LDS ("LDS-staged input tiles",
// BASELINE.json north_star) -- the bf16 form of nn.Conv2d / nn.ConvTranspose2d forward and of both data gradients
// (reference call sites: src/models.py:150,168,179 and their autograd).  gemm_gather.hip's kernel stages the gathered A
// operand once PER TAP (9x for a 3x3, 16x for a 4x4: the VGPR -> LDS path and 1.56x the algorithmic HBM traffic were its
// measured limits); here:
//
//   * a workgroup owns a 2-D tile of 4 rows x 32 columns of the output grid (128 pixels) x 128 output channels;
//   * per 64-channel chunk it stages the tile's input patch INCLUDING THE HALO once (6 x 34 pixels for a 3x3; 5 x 33 per
//     parity plane / parity class for the stride-2 forms, whose 16 taps are four 2x2 convolutions) -- global -> registers ->
//     LDS, double-buffered, loads issued a whole job ahead of the ds_write (a "job" = one patch + the taps that read it);
//   * every tap is then only a different LDS ROW ADDRESS: lane x of tile row y reads patch slot (y*PW + x) + delta(tap).
//     Slots are padded to 144 bytes (9 x 16): 16 consecutive slots land on 16 different 16-byte bank groups, so the
//     ds_read_b128 fragment reads are conflict-free with plain linear addressing (one v_add per tile row per tap);
//   * the weights never touch LDS: wave w owns output channels 32w..32w+31 for all 128 pixels, so its B-operand fragments
//     are 16-byte loads straight from the packed [tap][co][ci] image into registers (L2-resident, 288 KB for a 3x3),
//     double-buffered one tap ahead.  No barrier inside a job: one workgroup barrier per job (every 64-144 MFMAs per wave);
//   * v_mfma_f32_32x32x16_bf16 with the operands swapped (weights as A, pixels as B): an accumulator register group holds 4
//     consecutive channels of one pixel; the epilogue stages one tile row at a time (fp32) through the patch buffer that has
//     just been retired and stores 16-byte row-contiguous pieces with bias / ReLU / the ResBlock's skip-gradient add and
//     ReLU mask applied -- the same arithmetic, in the same order, as gemm_gather.hip's epilogue;
//   * workgroups walk tiles with a grid stride: patch and weight prefetch run across tile boundaries.
//
// MODE 1 (transposed 4/2/1 forward = data gradient of a 4/2/1 conv): the four output parity classes of one 4 x 32 block of
// the low-resolution grid are four consecutive "classes" of one workgroup (4 x 128 output pixels), each with its own 2x2
// taps, accumulators flushed per class.
#include "nsg_common.h"
#include <type_traits>

namespace {

// Diagnostics builds only (libnsg_diag.so, -DNSG_DIAG): run-time switches for A/B runs.  The product library has constants here.
NSG_DIAG_SWITCH(int, g_patch_gemm, 1)        // nsg_debug_set_patch_gemm: 0 sends everything back to gemm_gather.hip's kernel
NSG_DIAG_SWITCH(int, g_patch_grid_cap, 512)  // nsg_debug_set_patch_grid: workgroups per launch (2 per CU resident: each walks tiles with a grid stride); 0 = one tile each


// Workgroups along x for ntiles pixel tiles and ntn channel tiles (blockIdx.y): x * ntn workgroups are resident at once (two per
// CU), each walks its channel tile's pixel tiles with a grid stride.  (Round 2 capped x alone: with C_out = 256 the 1024
// workgroups ran as two batches of 512, 1.25 tiles each = four tile times for 2.5 tiles of work per slot; now three.)
inline int patch_grid(int ntiles, int ntn)
{
    const int cap = g_patch_grid_cap > 0 ? (g_patch_grid_cap / ntn > 0 ? g_patch_grid_cap / ntn : 1) : 0;
    return (cap > 0 && ntiles > cap) ? cap : ntiles;
}

constexpr int PG_MAX_JOBS = 16;
constexpr int PG_MAX_TAPS = 64;
constexpr int SLOT_BYTES = 16;      // LDS image of a patch chunk: [8 pieces of 16 bytes][slots]: a slot's pieces sit one PLANE apart

struct PatchJob {
    int c0_bytes;       // byte offset of the channel chunk inside a pixel
    short jy, jx;       // input pixel of patch slot (yy, xx) = (tile_y0 * sy + jy + yy * sy, tile_x0 * sx + jx + xx * sx)
    short oy, ox;       // MODE 1: parity class (py, px) of the outputs this job contributes to
    int flush;          // 1: the accumulators are complete after this job (store them)
};
struct PatchTap {
    int delta_bytes;        // LDS byte offset of the tap inside the patch (slot delta * SLOT_BYTES)
    unsigned woff_bytes;    // byte offset of (weight tap, chunk) in the fragment-ordered image: [tap][chunk][co / 32][4][64 lanes][8]
};

struct PatchGemmParams {
    const bf16_t *in;
    const bf16_t *w;
    const float *bias;
    bf16_t *out;
    const bf16_t *epi_add;
    const bf16_t *epi_mask;
    int B, IH, IW, CI, OH, OW, CO;
    int tiles_y, tiles_x, ntiles;
    FastDiv div_tx, div_tyx;
    int njobs, ntaps;
    int sy, sx;             // input step per patch slot (1, or 2 for the parity planes of a stride-2 conv)
    int os;                 // output step (1, or 2 for the transposed form)
    int flags;
    unsigned in_bytes, w_bytes;
    unsigned long long *stamps;     // diagnostics build only: [workgroup][8] cycle counts (never read by any kernel)
    float *stats;                   // STATS variant: [gridDim.y * gridDim.x][3][CO] (count, sum, M2 about the record's mean) of the stored output
    PatchJob jobs[PG_MAX_JOBS];
    PatchTap taps[PG_MAX_TAPS];
};

// PH x PW: patch extent in slots (6 x 34 or 5 x 33); NT: taps per job (9 or 4); ADD / MASK: the fused epilogue operands are
// compile-time (a run-time "maybe a load" makes hipcc guard every reuse of the destination registers with a conservative
// vmcnt that also waits for the tile's own STORES -- the epilogue then runs at store-acknowledge latency); STAMP: diagnostics.
// PRIV: the epilogue is WAVE-PRIVATE: wave w owns channels 32 w .. + 31 of all 128 pixels, so it turns its own accumulators
// into 64-byte output segments (4 lanes x 16 bytes per pixel) through a staging region of its own -- no workgroup barrier in
// the epilogue (the staged form has eight per tile), LDS writes / reads / stores of consecutive tile rows software-pipelined.
// Measured (same box, isolated launches on post-ReLU-like data, us): plain variants 196 / 323 / 367 wave-private against 198-202 /
// 327 / 374 staged (3x3, 4x4-s2, transposed); with fused operands the wave-private form LOSES (3x3 add + mask 236 vs 227,
// transposed + mask 464-475 vs 435-443: a lane's operand loads are 64-byte segments, two waves fetch every 128-byte line of
// the operand tensors), so those variants keep the staged epilogue: PRIV = plain variants only.  (Also measured at parity and
// removed again: pixel fragments two k-steps ahead, s_setprio / a start delay for the younger workgroup of a CU, the fused
// operands fetched in the job's last tap -- DESIGN.md 3.1e.)
// STATS (plain variant, wave-private epilogue): the batch statistics of the BatchNorm that follows, from the store phase -- a lane
// keeps the same 8 channels for every piece it stores, so it carries their running sums (about a pivot: its first value) across
// its tiles in registers; one (count, sum, M2) record per workgroup and channel tile at the end, pooled over the lanes in lane
// order (double).  Of the values AS STORED (rounded to bf16), real pixels only.  Replaces a read pass over the conv's output.
template <int PH, int PW, int NT, bool ADD, bool MASK, bool STAMP = false, bool STATS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void patch_gemm_kernel(const PatchGemmParams p)
{
    constexpr bool PRIV = !(ADD || MASK);
    constexpr int NSLOT = PH * PW;
    constexpr int NPIECE = NSLOT * 8;                       // 16-byte pieces of a patch (64 channels = 8 pieces per slot)
    constexpr int NP = (NPIECE + 255) / 256;                // pieces per thread
    // Prefetch distances.  vmcnt retires in order, so a wait for a tap's weight fragments (L2) also waits for every patch piece
    // (HBM / Infinity Cache) issued before them.  Round 2 fetched weights ONE tap ahead and wrote pieces to LDS two taps after
    // their fetch: a lone workgroup on a CU (its partner in an epilogue, or finished: the second-dispatched workgroup of a CU
    // loses the issue arbitration and runs 30 % longer) then ran no faster than two sharing the pipe -- its taps are 512 MFMA
    // cycles and the loads did not land in that time.  Now: NB weight buffers = fragments NB - 1 taps ahead; a piece is written
    // LD taps after its fetch.  The loop body is NB taps with static buffer roles (NB divides NT, LD divides NB).
    constexpr int NB = (NT == 9) ? 3 : 4;
    constexpr int LD = (NT == 9) ? 3 : 2;
    static_assert(NT % NB == 0 && NB % LD == 0 && NB >= 2, "static buffer roles inside a body of NB taps");
    constexpr int NLOAD = NT - LD;                          // taps that fetch patch pieces: 0 .. NLOAD - 1
    constexpr int PPT = (NP + NLOAD - 1) / NLOAD;           // pieces a thread fetches per tap
    // LDS image of a patch chunk: [piece 0..7][slot] x 16 bytes, PLANE a multiple of 256 bytes.  A fragment read takes ONE piece
    // of 16 (or 32) consecutive slots = consecutive 16-byte bank groups, whatever the tap's shift and for both MFMA shapes
    // (the lanes of one ds_read_b128 group that differ in piece sit whole PLANEs apart: same bank group as their slot alone).
    constexpr int NSLOT_PAD = (NSLOT + 15) / 16 * 16;
    constexpr int PLANE = NSLOT_PAD * 16;
    constexpr int BUF_BYTES = 8 * PLANE;
    constexpr int CPITCH = 132;                             // epilogue staging pitch (floats): [32 pixels][128 channels + 4]
    static_assert(32 * CPITCH * 4 <= BUF_BYTES, "a tile row of the output must fit a retired patch buffer");
    static_assert(PPT * NLOAD >= NP && NT >= 4, "every patch piece is fetched LD taps before the job ends");
    constexpr unsigned OOB = 0xfffffff0u;

    constexpr int STG_PITCH = 36;                           // PRIV staging: [32 pixels][32 channels + 4] floats per wave (conflict-free b128 writes)
    constexpr int STG_BYTES = 32 * STG_PITCH * 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][BUF_BYTES] (+ PRIV: [4 waves][STG_BYTES])

    // diagnostics (STAMP builds only): where a workgroup's cycles go -- tap loops / job boundaries / epilogues / the rest
    unsigned long long st_entry = 0, st_rt = 0, st_loop = 0, st_bound = 0, st_flush = 0, st_pro = 0, st_jobs = 0, st_t = 0;
    auto now = [&]() -> unsigned long long {
        if constexpr (STAMP) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            return t;
        }
        return 0ull;
    };
    if constexpr (STAMP) { st_entry = now(); st_rt = __builtin_amdgcn_s_memrealtime(); }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * 128;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.in), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w), 0, (int)p.w_bytes, 0x00020000);

    // ---- this thread's patch pieces: f = tid + 256 r -> slot (f & 7) + 8 (f >> 6), piece (f >> 3) & 7: eight consecutive lanes
    //      write eight consecutive slots of one piece plane (conflict-free ds_write_b128) and a wave-wide load still covers
    //      eight whole 128-byte pixel chunks ----
    const int slot0 = (tid & 7) + 8 * (tid >> 6);            // + 32 r
    const unsigned piece_b = ((tid >> 3) & 7) * 16;
    const unsigned pc_lds0 = ((tid >> 3) & 7) * PLANE + slot0 * 16;     // + r * 32 * 16

    // ---- fragment addressing ----
    // A operand of the MFMA = weights, read from the fragment-ordered image: (tap, chunk) block -> this wave's 32-channel
    // block -> four 1 KiB fragments -> lane: 16 bytes each, so one wave-wide load is 1 KiB contiguous.  (From the plain
    // [tap][co][ci] image the same load touches 32 cache lines for 32 bytes each and the step ran at the vector L1's line
    // rate: SQ counters, DESIGN.md.)  Fragment kk, lane (r = x31, h) = w[co = 32 w + r][ci = 16 kk + 8 h .. + 7].
    const unsigned wlane = (unsigned)((((n0 >> 5) + wave) * 4 * 64 + lane) * 16);
    // B operand = pixels: lane (x31, h) of tile row y reads slot (y PW + x31) + delta, piece 2 kk + h.
    unsigned arow[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) arow[y] = (unsigned)((y * PW + x31) * 16 + h * PLANE);

    // epilogue: this thread's 8-channel group and its bias.  Staged form: pixel (tid >> 4) + 16 e2 of the tile row, channels
    // 8 cg .. + 7.  PRIV: pixel ep + 16 e2, channels 32 wave + 8 eq .. + 7 (four lanes = 64 contiguous bytes of a pixel).
    const int cg = tid & 15;
    const int ep = lane >> 2, eq = lane & 3;
    const int ech = PRIV ? 32 * wave + 8 * eq : 8 * cg;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = p.bias ? p.bias[n0 + ech + e] : 0.f;
    const bool relu_out = (p.flags & NSG_RELU_OUT) != 0;
    static_assert(!STATS || (PRIV && !ADD && !MASK), "statistics ride in the plain variant's wave-private epilogue");
    float st_sum[STATS ? 8 : 1], st_sq[STATS ? 8 : 1], st_pv[STATS ? 8 : 1], st_cnt = 0.f;
#pragma unroll
    for (int e = 0; e < (STATS ? 8 : 1); ++e) { st_sum[e] = 0.f; st_sq[e] = 0.f; st_pv[e] = 0.f; }

    // The tap table lives in two VGPRs (lane q = entry q) and is read back with v_readlane: a scalar load inside the tap loop
    // would share lgkmcnt with the fragment reads and drain them.
    const int tab_delta = p.taps[lane].delta_bytes;
    const int tab_woff = (int)p.taps[lane].woff_bytes;
    static_assert(PG_MAX_TAPS <= 64, "one lane per table entry");
    auto tap_delta = [&](int q) { return __builtin_amdgcn_readlane(tab_delta, q); };
    auto tap_woff = [&](int q) { return __builtin_amdgcn_readlane(tab_woff, q); };
    // the job table the same way (lane j = job j): a run-time index into the by-value parameter struct from inside the job
    // lambdas made hipcc copy the whole struct to scratch
    const PatchJob myjob = p.jobs[lane & (PG_MAX_JOBS - 1)];
    const int tabj_c0 = myjob.c0_bytes;
    const int tabj_j = ((int)myjob.jy & 0xffff) | ((int)myjob.jx << 16);
    const int tabj_o = (int)myjob.oy | ((int)myjob.ox << 8) | (myjob.flush << 16);
    auto job_entry = [&](int j) {
        const int c0 = __builtin_amdgcn_readlane(tabj_c0, j), jj = __builtin_amdgcn_readlane(tabj_j, j), oo = __builtin_amdgcn_readlane(tabj_o, j);
        return PatchJob{c0, (short)(jj & 0xffff), (short)(jj >> 16), (short)(oo & 0xff), (short)((oo >> 8) & 0xff), (oo >> 16) & 1};
    };

    v16f acc[4];             // one 32 x 32 tile (32 channels x 32 pixels) per tile row
    v4f breg[NB][4];         // weight fragments of NB taps
    v4f afr[4];              // pixel fragments of one k-step: each is re-read for the next k-step right after its MFMA
    v4f ptmp[LD][PPT];       // patch pieces in flight: fetched in tap t (slot t % LD), written to LDS in tap t + LD

    // ---- job stream: (tile, job) for this workgroup, tiles bid, bid + grid, ... ----
    const int G = gridDim.x;
    struct Cursor { int tile, job; };
    auto advance = [&](Cursor &c) {
        if (++c.job == p.njobs) { c.job = 0; c.tile += G; }
    };
    auto tile_origin = [&](int tile, int &b, int &ty, int &tx) {
        b = nsg_div(tile, p.div_tyx);
        const int rem = tile - b * (p.tiles_y * p.tiles_x);
        ty = nsg_div(rem, p.div_tx);
        tx = rem - ty * p.tiles_x;
    };

    // The NEXT job's patch is fetched piece by piece inside the current job's taps: global -> a register in tap t, register ->
    // the OTHER patch buffer in tap t + 2 (that buffer is free: every wave passed the barrier that ended the job reading it).
    // vmcnt retires in order, so the piece loads are issued right AFTER a tap's weight loads: a weight wait never includes a
    // patch piece (HBM / Infinity Cache latency) issued less than two taps earlier.
    int pt_ok = 0, pt_iy0 = 0, pt_ix0 = 0;
    unsigned pt_base = 0;
    auto patch_origin = [&](const Cursor &c) {
        pt_ok = c.tile < p.ntiles;
        int b, ty, tx;
        tile_origin(pt_ok ? c.tile : 0, b, ty, tx);
        const PatchJob jb = job_entry(c.job);
        pt_iy0 = ty * 4 * p.sy + jb.jy;
        pt_ix0 = tx * 32 * p.sx + jb.jx;
        pt_base = (unsigned)(b * p.IH * p.IW) * (unsigned)p.CI * 2u + (unsigned)jb.c0_bytes;     // (< 4 GiB: the launcher checks)
    };
    auto load_piece = [&](int r) -> v4f {    // r: uniform, may be a run-time value
        const int slot = slot0 + 32 * r;
        const int yy = slot / PW, xx = slot - yy * PW;
        const int iy = pt_iy0 + yy * p.sy, ix = pt_ix0 + xx * p.sx;
        const bool ok = (slot < NSLOT) & (iy >= 0) & (iy < p.IH) & (ix >= 0) & (ix < p.IW) & (pt_ok != 0);
        const unsigned off = ok ? pt_base + (unsigned)((iy * p.IW + ix) * p.CI) * 2u + piece_b : OOB;
        return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)off, 0, 0));
    };
    auto store_piece = [&](int buf, int r, const v4f v) {
        if (slot0 + 32 * r < NSLOT) *reinterpret_cast<v4f *>(smem + buf * BUF_BYTES + pc_lds0 + r * 32 * 16) = v;
    };
    // tap t of a job reading buffer `buf` (t: run-time, uniform; HALF = t & 1 at compile time)
    auto patch_traffic = [&](auto HALF, int t, int buf) __attribute__((always_inline)) {
        constexpr int hf = decltype(HALF)::value;
        if (t >= LD) {
#pragma unroll
            for (int i = 0; i < PPT; ++i)
                if ((t - LD) * PPT + i < NP) store_piece(buf ^ 1, (t - LD) * PPT + i, ptmp[hf][i]);
        }
        if (t < NLOAD) {
#pragma unroll
            for (int i = 0; i < PPT; ++i)
                if (t * PPT + i < NP) ptmp[hf][i] = load_piece(t * PPT + i);
        }
    };

    auto load_b = [&](v4f (&bq)[4], int q) {
        const int so = tap_woff(q);
#if defined(NSG_DIAG) && defined(NSG_PATCH_FAKEW)     /* timing experiment (WRONG results, diagnostics builds only): a quarter of the weight-fragment traffic */
        bq[0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)wlane, so, 0));
        bq[1] = bq[0]; bq[2] = bq[0]; bq[3] = bq[0];
        return;
#endif
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            bq[kk] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wlane + 1024 * kk), so, 0));
    };
    auto read_a = [&](int y, int buf, int q, int kk) {
        afr[y] = *reinterpret_cast<const v4f *>(smem + buf * BUF_BYTES + tap_delta(q) + 2 * kk * PLANE + arow[y]);
    };
    auto read_a_first = [&](int buf, int q) {
#pragma unroll
        for (int y = 0; y < 4; ++y) read_a(y, buf, q, 0);
    };
    // One tap.  On entry afr holds the first k-step of tap q; on exit of tap qn.  At the end of a job qn = q: the next patch is
    // not visible before the job-boundary barrier, so the reads are dummies (branch-free) and run_job reads the real fragments
    // after the barrier.
    auto mfma_chain = [&](const v4f (&bq)[4], int buf, int q, int qn) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[kk]), __builtin_bit_cast(bf16x8, afr[y]), acc[y], 0, 0, 0);
                if (kk < 3) read_a(y, buf, q, kk + 1);
                else read_a(y, buf, qn, 0);
            }
            // pin the interleave: MFMA, then the read that refills its operand register (it lands under the next three MFMAs)
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };
    // A tap: the weight fragments of the tap after it (a whole tap to land), this tap's share of the patch traffic, the MFMAs.
    const unsigned out_bytes = (unsigned)(p.B * p.OH * p.OW) * (unsigned)p.CO * 2u;          // (< 4 GiB: the launcher checks)
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(ADD ? p.epi_add : p.out), 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mask = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(MASK ? p.epi_mask : p.out), 0, (int)out_bytes, 0x00020000);
    // epilogue state: byte offsets of this lane's 8 output pieces (tile row y, pixel ep + 16 e2; out of range = dropped / zero)
    // and the fused operands fetched for them
    unsigned eoff[4][2];
    v4f addv[ADD ? 4 : 1][2], maskv[MASK ? 4 : 1][2];
    auto epi_offsets = [&](int tile, const PatchJob &jb, bool live) {
        int b, ty, tx;
        tile_origin(tile, b, ty, tx);
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const int oy = (ty * 4 + y) * p.os + jb.oy;
                const int ox = (tx * 32 + (PRIV ? ep : (tid >> 4)) + 16 * e2) * p.os + jb.ox;
                eoff[y][e2] = (live && oy < p.OH && ox < p.OW) ? (unsigned)(((b * p.OH + oy) * p.OW + ox) * p.CO + n0 + ech) * 2u : OOB;
            }
    };
    auto epi_prefetch = [&]() {
        if constexpr (ADD) {    // the skip-path gradient (the ResBlock's residual add, backward)
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) addv[y][e2] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_add, (int)eoff[y][e2], 0, 0));
        }
        if constexpr (MASK) {   // the ReLU whose OUTPUT is epi_mask
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) maskv[y][e2] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_mask, (int)eoff[y][e2], 0, 0));
        }
    };
    // J (compile time): position of the tap inside the body of NB taps = its weight buffer; qb: table entry of the tap NB - 1 later
    auto tap = [&](auto JJ, int buf, int t, int q, int qn, int qb) __attribute__((always_inline)) {
        constexpr int J = decltype(JJ)::value;
        const v4f (&bq)[4] = breg[J];
        load_b(breg[(J + NB - 1) % NB], qb);
        __builtin_amdgcn_sched_barrier(0);      // (left alone, the scheduler sinks the loads to their first use)
        patch_traffic(std::integral_constant<int, J % LD>{}, t, buf);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chain(bq, buf, q, qn);
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[y][r] = 0.f;
    };

    // ---- epilogue of one output class of one tile: four passes (one tile row each) through the retired patch buffer; the
    //      fused operands of all four passes are fetched before the first.  Branch-free on purpose: output and fused operands
    //      go through buffer descriptors (an out-of-range offset drops the store / reads zero), because at every branch join
    //      hipcc's wait-count model turns pessimistic and guards the next register reuse with vmcnt(0) -- which also waits for
    //      the tile's own STORES: the epilogue then ran at store-acknowledge latency (8 K cycles per tile). ----
    // one output piece: 8 fp32 values -> bias, ReLU for the consumer, skip-gradient add, ReLU mask, bf16, 16-byte store
    auto epi_piece = [&](const v4f t0, const v4f t1, int y, int e2) __attribute__((always_inline)) {
        float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = relu_out ? fmaxf(v[e], 0.f) : v[e];
        if constexpr (ADD) {
            float t[8];
            Elem<bf16_t>::unpack16(addv[y][e2], t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        if constexpr (MASK) {
            float t[8];
            Elem<bf16_t>::unpack16(maskv[y][e2], t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = t[e] > 0.f ? v[e] : 0.f;
        }
        unsigned u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = nsg_pack_bf16(v[2 * i], v[2 * i + 1]);
        if constexpr (STATS) {
            const float live = eoff[y][e2] != OOB ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y0 = nsg_bitsf(u[i] << 16), y1 = nsg_bitsf(u[i] & 0xffff0000u);
                st_pv[2 * i] = (st_cnt == 0.f) ? y0 : st_pv[2 * i];
                st_pv[2 * i + 1] = (st_cnt == 0.f) ? y1 : st_pv[2 * i + 1];
                const float d0 = (y0 - st_pv[2 * i]) * live, d1 = (y1 - st_pv[2 * i + 1]) * live;
                st_sum[2 * i] += d0;
                st_sum[2 * i + 1] += d1;
                st_sq[2 * i] = __builtin_fmaf(d0, d0, st_sq[2 * i]);
                st_sq[2 * i + 1] = __builtin_fmaf(d1, d1, st_sq[2 * i + 1]);
            }
            st_cnt += live;
        }
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(u4{u[0], u[1], u[2], u[3]}, rs_out, (int)eoff[y][e2], 0, 0);
    };
    auto flush = [&](int tile, const PatchJob &jb, int buf) __attribute__((always_inline)) {
        if constexpr (PRIV) {
            // Wave-private: no barrier.  Tile row y: the wave's 32 pixels x 32 channels go to its staging region in the MFMA
            // layout (register group g of lane (x, h) = channels 8 g + 4 h .. + 3 of pixel x) and come back as 8 consecutive
            // channels of pixel ep + 16 e2 per lane.  LDS executes one wave's instructions in order, so row y + 1 may be
            // written right after row y's reads are ISSUED: write(y + 1) and read(y + 1) run under the arithmetic and the
            // stores of row y.
            epi_offsets(tile, jb, true);
            float *Sw = reinterpret_cast<float *>(smem + 2 * BUF_BYTES + wave * STG_BYTES);
            float *wdst = Sw + x31 * STG_PITCH + 4 * h;
            const float *rsrc = Sw + ep * STG_PITCH + 8 * eq;
            auto put_row = [&](int y) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<v4f *>(wdst + 8 * g) = v4f{acc[y][4 * g], acc[y][4 * g + 1], acc[y][4 * g + 2], acc[y][4 * g + 3]};
                __builtin_amdgcn_wave_barrier();
            };
            v4f rd[2][2][2];        // [row parity][e2][half of the 8 channels]
            auto get_row = [&](int y) {
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    rd[y & 1][e2][0] = *reinterpret_cast<const v4f *>(rsrc + 16 * e2 * STG_PITCH);
                    rd[y & 1][e2][1] = *reinterpret_cast<const v4f *>(rsrc + 16 * e2 * STG_PITCH + 4);
                }
                __builtin_amdgcn_wave_barrier();
            };
            put_row(0);
            get_row(0);
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                if (y < 3) { put_row(y + 1); get_row(y + 1); }
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) epi_piece(rd[y & 1][e2][0], rd[y & 1][e2][1], y, e2);
            }
            return;
        }
        // ---- staged form: four passes (one tile row each) through the retired patch buffer, two barriers each ----
        epi_offsets(tile, jb, true);
        epi_prefetch();
        float *Cs = reinterpret_cast<float *>(smem + buf * BUF_BYTES);
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            // this wave's 32 pixels x 32 channels of tile row y: register group g of lane (x, h) = channels 32 w + 8 g + 4 h .. + 3
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float *dst = Cs + x31 * CPITCH + 32 * wave + 8 * g + 4 * h;
                *reinterpret_cast<v4f *>(dst) = v4f{acc[y][4 * g], acc[y][4 * g + 1], acc[y][4 * g + 2], acc[y][4 * g + 3]};
            }
            __syncthreads();
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const int px = (tid >> 4) + 16 * e2;
                epi_piece(*reinterpret_cast<const v4f *>(Cs + px * CPITCH + 8 * cg), *reinterpret_cast<const v4f *>(Cs + px * CPITCH + 8 * cg + 4), y, e2);
            }
            __syncthreads();        // the staging rows are rewritten by the next pass (or by the next patch)
        }
    };

    // ---- the stream ----
    // Workgroups are dealt round-robin over the 8 XCDs (observed, speed only): give each XCD a contiguous run of tiles so that
    // the halo rows a tile shares with its vertical neighbours are hits in that XCD's L2.  Bijective for any grid size.
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, qq = G >> 3, rr = G & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    Cursor cur = {bid, 0};                  // job being computed
    Cursor pre = cur;                       // job whose patch is being fetched (one ahead of cur inside the loop)
    if (cur.tile >= p.ntiles) return;
    patch_origin(pre);
    for (int r = 0; r < NP; ++r) store_piece(0, r, load_piece(r));      // the first patch: nothing to overlap it with
#pragma unroll
    for (int j = 0; j + 1 < NB; ++j) load_b(breg[j], j);      // the first NB - 1 taps' weights (a job has at least NB taps)
    zero_acc();
    __syncthreads();
    int buf = 0;
    // One job: NT taps from patch buffer `buf`, in bodies of NB taps; all bodies but the last in a rolled loop, the last one
    // (whose final tap has no successor inside the job) at a static code position.
    auto run_job = [&](int qbase, int qnext_base) __attribute__((always_inline)) {
        using NO = std::integral_constant<bool, false>;
        using YES = std::integral_constant<bool, true>;
        auto tq = [&](int t) { return t < NT ? qbase + t : qnext_base + (t - NT); };     // table entry of tap t of the stream from this job on
        auto body = [&](int t0, auto LASTBODY) __attribute__((always_inline)) {
            constexpr bool LB = decltype(LASTBODY)::value;
#define NSG_TAP(J)                                                                                                              \
            if constexpr (J < NB) {                                                                                             \
                constexpr int NEXT = (LB && J == NB - 1) ? 0 : 1;                                                               \
                tap(std::integral_constant<int, J>{}, buf, t0 + J, qbase + t0 + J, qbase + t0 + J + NEXT, tq(t0 + J + NB - 1)); \
            }
            NSG_TAP(0) NSG_TAP(1) NSG_TAP(2) NSG_TAP(3)
#undef NSG_TAP
        };
        read_a_first(buf, qbase);
#pragma unroll 1
        for (int t0 = 0; t0 < NT - NB; t0 += NB) body(t0, NO{});
        body(NT - NB, YES{});
    };
    if constexpr (STAMP) st_pro = now() - st_entry;
    // One job of the stream (NB divides NT: every job starts in weight buffer 0, the roles are static per code position)
    auto do_job = [&]() __attribute__((always_inline)) {
        if constexpr (STAMP) st_t = now();
        const PatchJob jb = job_entry(cur.job);
        advance(pre);
        patch_origin(pre);                  // its pieces are fetched inside this job's taps
        const int qbase = cur.job * NT;
        const int qnext = (cur.job + 1 == p.njobs ? 0 : cur.job + 1) * NT;
        run_job(qbase, qnext);
        if constexpr (STAMP) { const unsigned long long t = now(); st_loop += t - st_t; st_t = t; st_jobs += 1; }
        __syncthreads();                    // job boundary: every wave is done with `buf`; the next patch (other buffer) is complete
        if constexpr (STAMP) { const unsigned long long t = now(); st_bound += t - st_t; st_t = t; }
        if (jb.flush) {
            flush(cur.tile, jb, buf);
            zero_acc();
            if constexpr (STAMP) { const unsigned long long t = now(); st_flush += t - st_t; st_t = t; }
        }
        advance(cur);
        buf ^= 1;
    };
    for (;;) {
        if (cur.tile >= p.ntiles) break;
        do_job();
    }
    if constexpr (STATS) {
        // (count, mean, M2) of each lane, pooled over the 16 lanes that share a channel octet (same wave, same eq; ep = 0 .. 15)
        // in ep order, in double: one record per workgroup.  The patch buffers are free: every wave is past the last job.
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem);      // [17][256]: 8 means, 8 M2s, the count
        const float inv = st_cnt > 0.f ? 1.f / st_cnt : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[e * 256 + tid] = st_pv[e] + st_sum[e] * inv;
            red[(8 + e) * 256 + tid] = fmaxf(st_sq[e] - st_sum[e] * st_sum[e] * inv, 0.f);
        }
        red[16 * 256 + tid] = st_cnt;
        __syncthreads();
        if (tid < 128) {
            const int ch = tid, wv = ch >> 5, eqq = (ch >> 3) & 3, e = ch & 7;      // channel ch of the tile: wave wv, octet eqq, element e
            double N = 0.0, S = 0.0;
            for (int r = 0; r < 16; ++r) {
                const int t = wv * 64 + r * 4 + eqq;
                const double n = red[16 * 256 + t];
                N += n;
                S += n * (double)red[e * 256 + t];
            }
            const double mu = N > 0.0 ? S / N : 0.0;
            double Q = 0.0;
            for (int r = 0; r < 16; ++r) {
                const int t = wv * 64 + r * 4 + eqq;
                const double n = red[16 * 256 + t];
                const double dl = (double)red[e * 256 + t] - mu;
                Q += n > 0.0 ? (double)red[(8 + e) * 256 + t] + n * dl * dl : 0.0;
            }
            float *dst = p.stats + (size_t)blockIdx.x * 3 * p.CO + n0 + ch;       // records of one workgroup index x: all channel tiles side by side
            dst[0] = (float)N;
            dst[p.CO] = (float)S;
            dst[2 * p.CO] = (float)Q;
        }
    }
    if constexpr (STAMP) {
        if (tid == 0 && p.stamps) {
            unsigned long long *o = p.stamps + 8 * (size_t)(blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = now() - st_entry; o[1] = __builtin_amdgcn_s_memrealtime() - st_rt; o[2] = st_loop; o[3] = st_bound;
            o[4] = st_flush; o[5] = st_pro; o[6] = st_jobs;
#ifdef NSG_PATCH_ABS_STAMPS
            o[3] = st_rt; o[5] = __builtin_amdgcn_s_memrealtime();      // absolute start / end (100 MHz ticks) instead of the boundary / prologue shares
#endif
            o[7] = ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);   // XCC_ID, HW_ID: which workgroups share a CU
        }
    }
}

template <int PH, int PW, int NT, bool ADD, bool MASK, bool STAMP, bool STATS = false>
int launch_patch(const PatchGemmParams &p, int ntiles_n, hipStream_t s)
{
    constexpr int NSLOT = PH * PW;
    constexpr size_t BUF_BYTES = (size_t)8 * ((NSLOT + 15) / 16 * 16) * 16;
    constexpr bool PRIV = !(ADD || MASK);
    const size_t lds = 2 * BUF_BYTES + (PRIV ? 4 * 32 * 36 * 4 : 0);     // + the four waves' private epilogue staging
    static LdsOptIn once;
    if (lds > 65536) {
        const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&patch_gemm_kernel<PH, PW, NT, ADD, MASK, STAMP, STATS>)}, lds, "patch_gemm");
        if (rc != NSG_OK) return rc;
    }
    // two workgroups per CU resident; more tiles than that are walked with a grid stride
    const int gx = patch_grid(p.ntiles, ntiles_n);
    hipLaunchKernelGGL((patch_gemm_kernel<PH, PW, NT, ADD, MASK, STAMP, STATS>), dim3((unsigned)gx, (unsigned)ntiles_n), dim3(256), lds, s, p);
    return nsg_check_launch("patch_gemm");
}

template <int PH, int PW, int NT, bool STAMP>
int launch_patch_epi(const PatchGemmParams &p, int ntiles_n, hipStream_t s)
{
    if constexpr (!STAMP) {
        if (p.stats) return launch_patch<PH, PW, NT, false, false, false, true>(p, ntiles_n, s);
    }
    if (p.epi_add && p.epi_mask) return launch_patch<PH, PW, NT, true, true, STAMP>(p, ntiles_n, s);
    if (p.epi_add) return launch_patch<PH, PW, NT, true, false, STAMP>(p, ntiles_n, s);
    if (p.epi_mask) return launch_patch<PH, PW, NT, false, true, STAMP>(p, ntiles_n, s);
    return launch_patch<PH, PW, NT, false, false, STAMP>(p, ntiles_n, s);
}


}  // namespace

#ifdef NSG_DIAG
extern "C" NSG_API void nsg_debug_set_patch_gemm(int on) { g_patch_gemm = on; }
extern "C" NSG_API void nsg_debug_set_patch_grid(int cap) { g_patch_grid_cap = cap; }
#endif

// kind of the launch for this file (0: 3x3 stride 1, 1: 4x4 stride 2 pad 1, 2: transposed 4/2/1), or -1: not taken
static int patch_kind(const GatherGemmParams &g)
{
    if (!g_patch_gemm) return -1;
    if (g.in_dtype != NSG_BF16 || g.out_dtype != NSG_BF16) return -1;
    if (g.flags & (NSG_RELU_IN | NSG_TANH_OUT)) return -1;
    // BatchNorm statistics from the store phase: the plain variant's wave-private epilogue only (else gemm_gather.hip's epilogue)
    if (g.stats && (g.epi_add || g.epi_mask || (g.flags & NSG_RELU_OUT) || g.stamps)) return -1;
    if (g.CI % 64 != 0 || g.CO % 128 != 0) return -1;
    const int chunks = g.CI / 64;
    int kind;
    if (g.mode == 1) kind = 2;
    else if (g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad >= 0 && g.pad <= 2 && g.pad_w >= 0 && g.pad_w <= 2) kind = 0;
    else if (g.KH == 4 && g.KW == 4 && g.stride == 2 && g.pad == 1 && g.pad_w == 1) kind = 1;
    else return -1;
    const int njobs = kind == 0 ? chunks : 4 * chunks;
    const int ntaps = kind == 0 ? 9 : 4;
    if (njobs > PG_MAX_JOBS || njobs * ntaps > PG_MAX_TAPS) return -1;
    if ((int64_t)g.B * nsg_cdiv(g.RH, 4) * nsg_cdiv(g.RW, 32) > 0x3fffffff) return -1;
    return kind;
}

// Statistics records a launch with these parameters writes when this file runs it (one per workgroup index x), or 0 when it does not.
int nsg_patch_gemm_stat_records(const GatherGemmParams &g)
{
    if (patch_kind(g) < 0) return 0;
    return patch_grid((int)((int64_t)g.B * nsg_cdiv(g.RH, 4) * nsg_cdiv(g.RW, 32)), g.CO / 128);
}

// Runs the launch on the patch-staged kernel when its shape is one this file implements; *handled says whether it did.
int nsg_launch_patch_gemm(const GatherGemmParams &g, hipStream_t s, bool *handled)
{
    *handled = false;
    const int kind = patch_kind(g);
    if (kind < 0) return NSG_OK;
    const int chunks = g.CI / 64;
    const int njobs = kind == 0 ? chunks : 4 * chunks;
    const int ntaps = kind == 0 ? 9 : 4;

    PatchGemmParams p = {};
    p.in = reinterpret_cast<const bf16_t *>(g.in);
    const int64_t wtaps = g.mode == 0 ? g.KH * g.KW : 16;
    const int64_t image = wtaps * g.CO * g.CI;                      // elements of the plain image; the fragment-ordered twin follows it
    p.w = reinterpret_cast<const bf16_t *>(g.w) + image;
    p.bias = g.bias;
    p.out = reinterpret_cast<bf16_t *>(g.out);
    p.epi_add = reinterpret_cast<const bf16_t *>(g.epi_add);
    p.epi_mask = reinterpret_cast<const bf16_t *>(g.epi_mask);
    p.B = g.B; p.IH = g.IH; p.IW = g.IW; p.CI = g.CI; p.OH = g.OH; p.OW = g.OW; p.CO = g.CO;
    constexpr int tcols = 32;                           // columns of a tile
    constexpr int pw3 = tcols + 2, pw2 = tcols + 1;     // patch widths in slots: 3x3 (halo 1 + 1), the stride-2 forms' parity planes / classes
    p.tiles_y = (int)nsg_cdiv(g.RH, 4);
    p.tiles_x = (int)nsg_cdiv(g.RW, tcols);
    const int64_t nt = (int64_t)g.B * p.tiles_y * p.tiles_x;
    if (nt > 0x3fffffff) return NSG_OK;
    p.ntiles = (int)nt;
    p.div_tx = nsg_fastdiv((uint32_t)p.tiles_x);
    p.div_tyx = nsg_fastdiv((uint32_t)p.tiles_y * (uint32_t)p.tiles_x);
    p.njobs = njobs; p.ntaps = ntaps;
    p.sy = p.sx = kind == 1 ? 2 : 1;
    p.os = kind == 2 ? 2 : 1;
    p.flags = g.flags;
    p.in_bytes = g.in_bytes; p.w_bytes = g.w_bytes;
    p.stamps = g.stamps;
    p.stats = g.stats;
    const int64_t blk = (int64_t)(g.CO / 32) * 4096;    // bytes of one (tap, 64-channel chunk) block of the fragment-ordered image
    auto wblock = [&](int ws, int c) { return (unsigned)(((int64_t)ws * chunks + c) * blk); };
    int j = 0;
    if (kind == 0) {
        for (int c = 0; c < chunks; ++c, ++j) {
            p.jobs[j] = PatchJob{c * 128, (short)-g.pad, (short)-g.pad_w, 0, 0, c + 1 == chunks};
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw)
                    p.taps[j * 9 + kh * 3 + kw] = PatchTap{(kh * pw3 + kw) * SLOT_BYTES, wblock(kh * 3 + kw, c)};
        }
    } else if (kind == 1) {
        for (int c = 0; c < chunks; ++c)
            for (int pl = 0; pl < 4; ++pl, ++j) {
                const int ph = pl >> 1, pw = pl & 1;
                p.jobs[j] = PatchJob{c * 128, (short)(ph - 1), (short)(pw - 1), 0, 0, j + 1 == njobs};
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b)
                        p.taps[j * 4 + a * 2 + b] = PatchTap{(a * pw2 + b) * SLOT_BYTES,
                                                           wblock((2 * a + ph) * 4 + (2 * b + pw), c)};
            }
    } else {
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            for (int c = 0; c < chunks; ++c, ++j) {
                p.jobs[j] = PatchJob{c * 128, (short)(py - 1), (short)(px - 1), (short)py, (short)px, c + 1 == chunks};
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b)
                        p.taps[j * 4 + a * 2 + b] = PatchTap{((1 - a) * pw2 + (1 - b)) * SLOT_BYTES,
                                                           wblock(((1 - py) + 2 * a) * 4 + (1 - px) + 2 * b, c)};
            }
        }
    }
    *handled = true;
    const int ntn = g.CO / 128;
#ifdef NSG_DIAG
    if (g.stamps) {     // diagnostics build of the same kernel (scripts/patch_gemm_shares.py)
        if (kind == 0) return launch_patch_epi<6, 34, 9, true>(p, ntn, s);
        return launch_patch_epi<5, 33, 4, true>(p, ntn, s);
    }
#endif
    if (kind == 0) return launch_patch_epi<6, 34, 9, false>(p, ntn, s);
    return launch_patch_epi<5, 33, 4, false>(p, ntn, s);
}
