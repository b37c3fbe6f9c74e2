// out[k][:] = sum over rows i with idx[i] == k of g[i][:]  as a SORTED SEGMENT SUM (index_add_, vector_quantization.py:60-61;
// autograd of torch.index_select, src/models.py:137; the per-code sums of the EMA extension and of the lean bf16 step).
//
// The one-hot GEMM of gemm_wgrad.hip does 2*N*K*D flops for N*D adds: fine at K = 512 (209 us on the bf16 pipe, 840 us in
// fp32), 0.63 ms / 3.2 ms at K = 8192.  Here the rows are put in (code, row) order by a stable counting sort of their INDICES
// and every code's rows are summed in that order, so the result is bitwise reproducible and only N*D*4 bytes move:
//   A  seg_hist      per block of 1024 rows: LDS histogram of the codes                    -> blockcnt[b][k]
//   B1 seg_scan_blk  per code: exclusive scan over blocks (in place), total[k]; counts[k]
//   B2 seg_scan_code one block: base[k] = exclusive scan of total; chunks of <= 128 rows per code: chunkbase[k]
//   C  seg_place     one wave per block, 64 rows at a time in row order: position = running[k] + the row's rank among the 64
//                    rows' lanes of its code, found through a lane mask per code built with LDS atomic ORs (order-independent)
//                                                                                           -> perm[position] = row
//   D  seg_sum       one wave per (code, chunk): fp32 sum of its rows in position order     -> partial[chunk][:]
//   E  seg_final     per code: partials in chunk order                                      -> out[k][:]
#include "nsg_common.h"

namespace {

constexpr int SEG_ROWS = 1024;      // rows per block of the counting sort
constexpr int SEG_CH = 128;         // rows per summation chunk

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

__global__ __launch_bounds__(256) void seg_scan_blk_kernel(int *__restrict__ blockcnt, int nb, int K, int *__restrict__ total, float *__restrict__ counts)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    int run = 0;
    int b = 0;
    for (; b + 8 <= nb; b += 8) {           // loads eight at a time (they are independent), the running sum in block order
        int c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = blockcnt[(size_t)(b + j) * K + k];
#pragma unroll
        for (int j = 0; j < 8; ++j) { blockcnt[(size_t)(b + j) * K + k] = run; run += c[j]; }
    }
    for (; b < nb; ++b) {
        const int c = blockcnt[(size_t)b * K + k];
        blockcnt[(size_t)b * K + k] = run;
        run += c;
    }
    total[k] = run;
    if (counts) counts[k] = (float)run;
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
__global__ __launch_bounds__(64) void seg_sum_kernel(const float *__restrict__ g, const int *__restrict__ perm, const int *__restrict__ total,
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
    const int p0 = base[k] + j * SEG_CH, p1 = min(base[k] + total[k], p0 + SEG_CH);
    const int lane = threadIdx.x;
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
        // combine the rpi sub-rows of each piece in sub order through LDS
        __shared__ v4f red[64];
        red[lane] = acc;
        __syncthreads();
        if (lane < ppr) {
            v4f s = red[lane];
            for (int q = 1; q < rpi; ++q) s += red[q * ppr + lane];
            *reinterpret_cast<v4f *>(partial + (size_t)chunk * D + lane * 4) = s;
        }
    } else {
        for (int pc = lane; pc < ppr; pc += 64) {        // wide rows: a lane walks its pieces
            v4f s = {0.f, 0.f, 0.f, 0.f};
            for (int p = p0; p < p1; ++p) s += *reinterpret_cast<const v4f *>(g + (size_t)perm[p] * D + pc * 4);
            *reinterpret_cast<v4f *>(partial + (size_t)chunk * D + pc * 4) = s;
        }
    }
}

__global__ __launch_bounds__(256) void seg_final_kernel(const float *__restrict__ partial, const int *__restrict__ chunkbase, int K, int D,
                                                        float *__restrict__ out)
{
    const int ppr = D >> 2;
    const int64_t totalp = (int64_t)K * ppr;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < totalp; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i / ppr), pc = (int)(i - (int64_t)k * ppr);
        const int c0 = chunkbase[k], c1 = chunkbase[k + 1];
        v4f s = {0.f, 0.f, 0.f, 0.f};
        for (int c = c0; c < c1; ++c) s += *reinterpret_cast<const v4f *>(partial + (size_t)c * D + pc * 4);
        *reinterpret_cast<v4f *>(out + (size_t)k * D + pc * 4) = s;
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
    hipLaunchKernelGGL(seg_scan_blk_kernel, dim3((K + 255) / 256), dim3(256), 0, s, blockcnt, L.nb, K, total, counts_out);
    hipLaunchKernelGGL(seg_scan_code_kernel, dim3(1), dim3(1024), 0, s, total, K, base, chunkbase);
    static LdsOptIn once;
    if ((size_t)K * 12 > 65536 - 1024) {
        const int rc = nsg_lds_opt_in(once, {reinterpret_cast<const void *>(&seg_place_kernel)}, 8192 * 12, "index_add_rows_sorted");
        if (rc != NSG_OK) return rc;
    }
    hipLaunchKernelGGL(seg_place_kernel, dim3(L.nb), dim3(64), (size_t)K * 12, s, idx, N, K, blockcnt, base, perm);
    hipLaunchKernelGGL(seg_sum_kernel, dim3((unsigned)L.maxchunks), dim3(64), 0, s, g, perm, total, base, chunkbase, K, D, partial);
    const int64_t nb = nsg_cdiv((int64_t)K * (D / 4), 256);
    hipLaunchKernelGGL(seg_final_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, s, partial, chunkbase, K, D, out);
    return nsg_check_launch("index_add_rows_sorted");
}

}  // extern "C"
