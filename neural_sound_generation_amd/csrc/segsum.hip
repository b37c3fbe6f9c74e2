// out[k][:] = sum over rows i with idx[i] == k of g[i][:]  as a SORTED SEGMENT SUM (index_add_, vector_quantization.py:60-61;
// autograd of torch.index_select, src/models.py:137; the per-code sums of the EMA extension and of the lean bf16 step).
//
// The one-hot GEMM of gemm_wgrad.hip does 2*N*K*D flops for N*D adds: fine at K = 512 (209 us on the bf16 pipe, 840 us in
// fp32), 0.63 ms / 3.2 ms at K = 8192.  Here the rows are put in (code, row) order by a stable counting sort of their INDICES
// and every code's rows are summed in that order, so the result is bitwise reproducible and only N*D*4 bytes move:
//   A  seg_hist      per block of 1024 rows: LDS histogram of the codes                    -> blockcnt[b][k]
//   B1 seg_scan_blk  per code: exclusive scan over blocks (in place), total[k]; counts[k]
//   B2 seg_scan_code one block: base[k] = exclusive scan of total; chunks of <= 512 rows per code: chunkbase[k]
//   C  seg_place     one wave per block, 64 rows at a time in row order: position = running[k] + the row's rank among the 64
//                    rows' lanes of its code, found through a lane mask per code built with LDS atomic ORs (order-independent)
//                                                                                           -> perm[position] = row
//   D  seg_sum       four waves per (code, chunk): fp32 sums of their 128 rows in position order, added in wave order -> partial[chunk][:]
//   E  seg_final     per code: partials in chunk order                                      -> out[k][:]
#include "nsg_common.h"

namespace {

constexpr int SEG_ROWS = 1024;      // rows per block of the counting sort
constexpr int SEG_CH = 512;         // rows per summation chunk: 4 waves x 128 rows

struct SegLayout {
    size_t blockcnt, total, base, chunkbase, perm, partial, bytes;
    int nb;
    int64_t maxchunks;
};
SegLayout seg_layout(int64_t N, int D, int K)
{
    SegLayout L;
    L.nb = (int)nsg_cdiv(N, SEG_ROWS);
    L.maxchunks = nsg_cdiv(N, SEG_CH) + K;
    size_t o = 0;
    L.blockcnt = o; o += nsg_align_up((size_t)L.nb * K * 4, 256);
    L.total = o; o += nsg_align_up((size_t)K * 4, 256);
    L.base = o; o += nsg_align_up((size_t)K * 4, 256);
    L.chunkbase = o; o += nsg_align_up((size_t)(K + 1) * 4, 256);
    L.perm = o; o += nsg_align_up((size_t)N * 4, 256);
    L.partial = o; o += nsg_align_up((size_t)L.maxchunks * D * 4, 256);
    L.bytes = o;
    return L;
}

__global__ __launch_bounds__(256) void seg_hist_kernel(const int64_t *__restrict__ idx, int64_t N, int K, int *__restrict__ blockcnt)
{
    extern __shared__ int bins[];
    for (int k = threadIdx.x; k < K; k += 256) bins[k] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * SEG_ROWS;
    for (int i = threadIdx.x; i < SEG_ROWS; i += 256) {
        const int64_t r = r0 + i;
        if (r < N) {
            const int64_t c = idx[r];
            if (c >= 0 && c < K) atomicAdd(&bins[(int)c], 1);       // (integer: the result does not depend on the order)
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) blockcnt[(size_t)blockIdx.x * K + k] = bins[k];
}

// blockcnt[b][k] -> its exclusive scan over b (in place), total[k], counts[k].  Block = 32 codes x 8 segments of the block range:
// a thread loads its whole segment at once (SEG_SCAN_MAX values in registers: two memory round trips per thread in all instead
// of one per 8 blocks), the 8 segment totals of a code meet in LDS.
constexpr int SEG_SCAN_MAX = 96;     // blocks per segment held in registers: nb <= 8 * 96 per pass (more: further passes carry the sum)
__global__ __launch_bounds__(256) void seg_scan_blk_kernel(int *__restrict__ blockcnt, int nb, int K, int *__restrict__ total, float *__restrict__ counts)
{
    __shared__ int segsum[8][32];
    const int kk = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + kk;
    const bool live = k < K;
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += 8 * SEG_SCAN_MAX) {
        const int span = min(nb - b0, 8 * SEG_SCAN_MAX);
        const int per = (span + 7) / 8;                    // blocks per segment in this pass (<= SEG_SCAN_MAX)
        const int s0 = b0 + seg * per, s1 = min(b0 + span, s0 + per);
        int c[SEG_SCAN_MAX];
#pragma unroll
        for (int j = 0; j < SEG_SCAN_MAX; ++j) c[j] = blockcnt[(size_t)min(s0 + j, nb - 1) * K + (live ? k : 0)];     // unconditional, clamped
#pragma unroll
        for (int j = 0; j < SEG_SCAN_MAX; ++j) c[j] &= -(int)(live && s0 + j < s1);     // (a select here lets hipcc sink every load into its own branch)
        int sum = 0;
#pragma unroll
        for (int j = 0; j < SEG_SCAN_MAX; ++j) sum += c[j];
        segsum[seg][kk] = sum;
        __syncthreads();
        int run = carry, all = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int v = segsum[q][kk]; run += q < seg ? v : 0; all += v; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SEG_SCAN_MAX; ++j) {
            if (live && s0 + j < s1) blockcnt[(size_t)(s0 + j) * K + k] = run;
            run += c[j];
        }
        carry += all;
    }
    if (live && seg == 0) {
        total[k] = carry;
        if (counts) counts[k] = (float)carry;
    }
}

// one block of 1024 threads: exclusive scans over the codes of total[] (-> base) and of ceil(total / SEG_CH) (-> chunkbase)
__global__ __launch_bounds__(1024) void seg_scan_code_kernel(const int *__restrict__ total, int K, int *__restrict__ base, int *__restrict__ chunkbase)
{
    __shared__ int sa[1024], sb[1024];
    const int tid = threadIdx.x;
    const int per = (K + 1023) / 1024;
    const int k0 = tid * per, k1 = min(K, k0 + per);
    int a = 0, b = 0;
    for (int k = k0; k < k1; ++k) { a += total[k]; b += (total[k] + SEG_CH - 1) / SEG_CH; }
    sa[tid] = a; sb[tid] = b;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan (integers: exact)
        const int va = tid >= off ? sa[tid - off] : 0, vb = tid >= off ? sb[tid - off] : 0;
        __syncthreads();
        sa[tid] += va; sb[tid] += vb;
        __syncthreads();
    }
    int ra = sa[tid] - a, rb = sb[tid] - b;              // exclusive prefix of this thread's run of codes
    for (int k = k0; k < k1; ++k) {
        base[k] = ra; chunkbase[k] = rb;
        ra += total[k]; rb += (total[k] + SEG_CH - 1) / SEG_CH;
    }
    if (tid == 1023) chunkbase[K] = sb[1023];
}

// one wave per block of SEG_ROWS rows: rows in order, 64 at a time.  The rows of one code inside the 64 find each other through
// a 64-bit lane mask per code built with LDS atomic ORs (commutative: no dependence on arrival order): rank = number of set
// bits below the lane, so positions go out in row order; the code's lowest lane advances its running position.
__global__ __launch_bounds__(64) void seg_place_kernel(const int64_t *__restrict__ idx, int64_t N, int K, const int *__restrict__ blockoff,
                                                       const int *__restrict__ base, int *__restrict__ perm)
{
    extern __shared__ int lds[];
    int *run = lds;                                          // [K] next position of each code
    unsigned *mlo = reinterpret_cast<unsigned *>(lds + K);   // [K] lanes 0-31 holding the code in the current 64 rows
    unsigned *mhi = mlo + K;                                 // [K] lanes 32-63
    const int lane = threadIdx.x;
    for (int k = lane; k < K; k += 64) { run[k] = base[k] + blockoff[(size_t)blockIdx.x * K + k]; mlo[k] = 0u; mhi[k] = 0u; }
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * SEG_ROWS;
    for (int i0 = 0; i0 < SEG_ROWS; i0 += 64) {
        if (r0 + i0 >= N) break;
        const int64_t r = r0 + i0 + lane;
        int c = -1;
        if (r < N) {
            const int64_t cc = idx[r];
            c = (cc >= 0 && cc < K) ? (int)cc : -1;          // rows with an index outside [0, K) contribute nothing
        }
        if (c >= 0) atomicOr(lane < 32 ? &mlo[c] : &mhi[c], 1u << (lane & 31));
        __syncthreads();
        unsigned lo = 0u, hi = 0u;
        if (c >= 0) { lo = mlo[c]; hi = mhi[c]; }
        __syncthreads();
        if (c >= 0) {
            const unsigned blo = lane < 32 ? lo & ((1u << lane) - 1u) : lo;
            const unsigned bhi = lane < 32 ? 0u : hi & ((1u << (lane - 32)) - 1u);
            const int rank = __popc(blo) + __popc(bhi);
            perm[run[c] + rank] = (int)r;
        }
        __syncthreads();
        if (c >= 0) {
            const bool lowest = lo ? (lane < 32 && (lo & ((1u << lane) - 1u)) == 0u) : (lane >= 32 && (hi & ((1u << (lane - 32)) - 1u)) == 0u);
            if (lowest) {
                run[c] += __popc(lo) + __popc(hi);
                mlo[c] = 0u;
                mhi[c] = 0u;
            }
        }
        __syncthreads();
    }
}

// one wave per chunk: rows perm[p0 .. p1) of code k summed in position order.  D = 4 * 64 * RPL... a lane owns one 16-byte piece
// of a row; a wave-wide load covers 1024 / (4 D) rows; lanes that own the same piece of different rows are combined in lane order.
__global__ __launch_bounds__(256) void seg_sum_kernel(const float *__restrict__ g, const int *__restrict__ perm, const int *__restrict__ total,
                                                     const int *__restrict__ base, const int *__restrict__ chunkbase, int K, int D,
                                                     float *__restrict__ partial)
{
    const int chunk = blockIdx.x;
    if (chunk >= chunkbase[K]) return;
    // binary search: the code k with chunkbase[k] <= chunk < chunkbase[k + 1]
    int lo = 0, hi = K;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunkbase[mid] <= chunk) lo = mid; else hi = mid;
    }
    const int k = lo;
    const int j = chunk - chunkbase[k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pend = min(base[k] + total[k], base[k] + (j + 1) * SEG_CH);
    const int p0 = min(pend, base[k] + j * SEG_CH + wave * (SEG_CH / 4)), p1 = min(pend, p0 + SEG_CH / 4);   // this wave's quarter
    const int ppr = D >> 2;                 // 16-byte pieces per row (D % 4 == 0)
    const int rpi = 64 / ppr;               // rows per wave-wide load when ppr <= 64
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    if (ppr <= 64) {
        const int sub = lane / ppr, pc = lane - sub * ppr;
        const bool act = sub < rpi;
        int p = p0;
        for (; p + 4 * rpi <= p1; p += 4 * rpi) {       // four loads in flight, added in position order
            v4f v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = act ? perm[p + u * rpi + sub] : 0;
                v[u] = *reinterpret_cast<const v4f *>(g + (size_t)row * D + pc * 4);
            }
            if (act) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += v[u];
            }
        }
        for (; p < p1; p += rpi) {
            if (act && p + sub < p1) acc += *reinterpret_cast<const v4f *>(g + (size_t)perm[p + sub] * D + pc * 4);
        }
        // combine the rpi sub-rows of each piece in sub order, then the four waves in wave order, through LDS
        __shared__ v4f red[256];
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < ppr) {
            v4f s = {0.f, 0.f, 0.f, 0.f};
            for (int w = 0; w < 4; ++w) {
                v4f sw = red[64 * w + threadIdx.x];
                for (int q = 1; q < rpi; ++q) sw += red[64 * w + q * ppr + threadIdx.x];
                s += sw;
            }
            *reinterpret_cast<v4f *>(partial + (size_t)chunk * D + threadIdx.x * 4) = s;
        }
    } else {
        for (int pc = threadIdx.x; pc < ppr; pc += 256) {        // wide rows: a thread walks its pieces over the whole chunk
            v4f s = {0.f, 0.f, 0.f, 0.f};
            for (int p = base[k] + j * SEG_CH; p < pend; ++p) s += *reinterpret_cast<const v4f *>(g + (size_t)perm[p] * D + pc * 4);
            *reinterpret_cast<v4f *>(partial + (size_t)chunk * D + pc * 4) = s;
        }
    }
}

// one block per code: 256 threads = (pieces of a row) x G groups; group q adds the chunks q, q + G, ... of the code in that order,
// eight loads in flight, and the G group sums are added in group order -- a fixed shape whatever the code's share of the rows (a
// collapsed codebook puts most rows, hence thousands of partials, on a few codes)
__global__ __launch_bounds__(256) void seg_final_kernel(const float *__restrict__ partial, const int *__restrict__ chunkbase, int K, int D,
                                                        float *__restrict__ out)
{
    __shared__ v4f red[256];
    const int k = blockIdx.x;
    const int ppr = D >> 2;
    const int c0 = chunkbase[k], c1 = chunkbase[k + 1];
    for (int pb = 0; pb < ppr; pb += 256) {                 // (D > 1024: several passes)
        const int P = min(ppr - pb, 256);                   // pieces in this pass: a power of two <= 64, or a multiple of 64
        const int G = P <= 64 ? 256 / P : 1;
        const int pc = pb + (int)threadIdx.x % P, q = (int)threadIdx.x / P;
        const bool act = (int)threadIdx.x < P * G;
        v4f s = {0.f, 0.f, 0.f, 0.f};
        int c = c0 + q;
        for (; c + 7 * G < c1; c += 8 * G) {
            v4f v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const v4f *>(partial + (size_t)(c + u * G) * D + (act ? pc : 0) * 4);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < c1; c += G) s += *reinterpret_cast<const v4f *>(partial + (size_t)c * D + (act ? pc : 0) * 4);
        red[threadIdx.x] = s;
        __syncthreads();
        if ((int)threadIdx.x < P) {
            v4f t = red[threadIdx.x];
            for (int g2 = 1; g2 < G; ++g2) t += red[g2 * P + threadIdx.x];
            *reinterpret_cast<v4f *>(out + (size_t)k * D + pc * 4) = t;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

size_t nsg_index_add_sorted_workspace_bytes(int64_t N, int32_t D, int32_t K)
{
    if (N <= 0 || D <= 0 || K <= 0) return 0;
    return seg_layout(N, D, K).bytes;
}

int nsg_index_add_rows_sorted(const int64_t *idx, const float *g, int64_t N, int32_t D, int32_t K, float *out, float *counts_out,
                              void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(idx && g && out && N > 0 && D > 0 && K > 0, NSG_E_INVALID, "nsg_index_add_rows_sorted: bad argument");
    NSG_REQUIRE(D % 4 == 0 && nsg_aligned16(g) && nsg_aligned16(out), NSG_E_UNSUPPORTED, "nsg_index_add_rows_sorted: D %% 4 == 0 and 16-byte aligned tensors");
    NSG_REQUIRE(N < 0x7fffffffll && K <= 8192 && 64 % (D / 4 > 64 ? 64 : D / 4) == 0, NSG_E_UNSUPPORTED,
                "nsg_index_add_rows_sorted: N < 2^31, K <= 8192, D a power of two up to 256 (or a multiple of 256)");
    const SegLayout L = seg_layout(N, D, K);
    NSG_REQUIRE(workspace && workspace_bytes >= L.bytes, NSG_E_WORKSPACE, "nsg_index_add_rows_sorted: workspace too small");
    char *ws = reinterpret_cast<char *>(workspace);
    int *blockcnt = reinterpret_cast<int *>(ws + L.blockcnt), *total = reinterpret_cast<int *>(ws + L.total);
    int *base = reinterpret_cast<int *>(ws + L.base), *chunkbase = reinterpret_cast<int *>(ws + L.chunkbase);
    int *perm = reinterpret_cast<int *>(ws + L.perm);
    float *partial = reinterpret_cast<float *>(ws + L.partial);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(seg_hist_kernel, dim3(L.nb), dim3(256), (size_t)K * 4, s, idx, N, K, blockcnt);
    hipLaunchKernelGGL(seg_scan_blk_kernel, dim3((K + 31) / 32), dim3(256), 0, s, blockcnt, L.nb, K, total, counts_out);
    hipLaunchKernelGGL(seg_scan_code_kernel, dim3(1), dim3(1024), 0, s, total, K, base, chunkbase);
    static LdsOptIn once;
    if ((size_t)K * 12 > 65536 - 1024) {
        const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&seg_place_kernel)}, 8192 * 12, "index_add_rows_sorted");
        if (rc != NSG_OK) return rc;
    }
    hipLaunchKernelGGL(seg_place_kernel, dim3(L.nb), dim3(64), (size_t)K * 12, s, idx, N, K, blockcnt, base, perm);
    hipLaunchKernelGGL(seg_sum_kernel, dim3((unsigned)L.maxchunks), dim3(256), 0, s, g, perm, total, base, chunkbase, K, D, partial);
    hipLaunchKernelGGL(seg_final_kernel, dim3((unsigned)K), dim3(256), 0, s, partial, chunkbase, K, D, out);
    return nsg_check_launch("index_add_rows_sorted");
}

}  // extern "C"
