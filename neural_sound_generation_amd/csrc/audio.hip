// Mel -> waveform inversion of the reference's epoch loop (src/main.py:164-197 -> src/audio_tacotron.py:99-116,142-153
// with use_lws=False): denormalise, dB -> amplitude, pseudo-inverse mel basis, power, Griffin-Lim, inverse pre-emphasis.
// SURVEY.md section 8f row 4.  The reference delegates the transforms to librosa (stft / istft) and scipy (lfilter);
// their published algorithms are restated here as HIP kernels (and in numpy in oracle/audio_oracle.py):
//   * radix-2 Stockham FFT of one frame per 256-thread workgroup, entirely in LDS (ping-pong buffers, twiddle table);
//   * stft: centred frames with reflect padding x periodic Hann window -> FFT -> keep 1 + N/2 bins; fused with the
//     Griffin-Lim phase update  spec = |S| * X / |X|;
//   * istft: Hermitian extension -> inverse FFT -> x window -> frame buffer; overlap-add as a gather (deterministic)
//     divided by the window's sum of squares, centre-trimmed;
//   * inverse pre-emphasis y[n] = x[n] + k y[n-1]: chunked, each chunk re-running a discarded warm-up (k^W < 1e-9).
// fp32 throughout (complex as float2).  These are LDS- / HBM-bound kernels; nothing here touches the matrix pipe.
#include "nsg_common.h"
#include <math.h>

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f cmul(v2f a, v2f b) { return v2f{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// In-LDS radix-2 Stockham FFT of N = 2^LOG2N points by 256 threads.  buf0 holds the input; returns the buffer holding the
// result.  tw[t] = exp(-2 pi i t / N), t < N/2.  inverse: conjugated twiddles (no 1/N scaling).
template <int LOG2N>
__device__ v2f *fft_lds(v2f *buf0, v2f *buf1, const v2f *tw, int tid, bool inverse)
{
    constexpr int N = 1 << LOG2N;
    v2f *in = buf0, *out = buf1;
#pragma unroll 1
    for (int s = 0; s < LOG2N; ++s) {
        const int Ns = 1 << s;
        __syncthreads();
        for (int j = tid; j < N / 2; j += 256) {
            const int k = j & (Ns - 1);
            v2f w = tw[k << (LOG2N - 1 - s)];
            if (inverse) w.y = -w.y;
            const v2f a = in[j];
            const v2f b = cmul(w, in[j + N / 2]);
            const int j0 = ((j >> s) << (s + 1)) + k;
            out[j0] = a + b;
            out[j0 + Ns] = a - b;
        }
        v2f *t = in; in = out; out = t;
    }
    __syncthreads();
    return in;
}

template <int LOG2N>
__device__ __forceinline__ void fill_twiddles(v2f *tw, int tid)
{
    constexpr int N = 1 << LOG2N;
    for (int t = tid; t < N / 2; t += 256) {
        float sn, cs;
        sincospif(-2.0f * (float)t / (float)N, &sn, &cs);
        tw[t] = v2f{cs, sn};
    }
}

// S[b][t][f] = max(1e-10, sum_m inv[f][m] * amp(mel[b][m][t]))^power,  amp = 10^((clip(mel,0,max_abs)*(-min_db)/max_abs + min_db + ref_db)/20)
__global__ __launch_bounds__(256) void mel_to_linear_kernel(const float *__restrict__ mel, const float *__restrict__ inv, float *__restrict__ S,
                                                            int B, int n_mels, int T, int F, float min_db, float ref_db, float max_abs, float power)
{
    extern __shared__ float amp[];     // [n_mels] of this (b, t)
    const int bt = blockIdx.x;
    const int b = bt / T, t = bt - b * T;
    for (int m = threadIdx.x; m < n_mels; m += 256) {
        const float v = fminf(fmaxf(mel[((size_t)b * n_mels + m) * T + t], 0.f), max_abs);
        const float db = v * (-min_db) / max_abs + min_db + ref_db;
        amp[m] = exp10f(db * 0.05f);
    }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 256) {
        float acc = 0.f;
        for (int m = 0; m < n_mels; ++m) acc = fmaf(inv[(size_t)f * n_mels + m], amp[m], acc);
        S[((size_t)b * T + t) * F + f] = powf(fmaxf(acc, 1e-10f), power);
    }
}

// spec[b][t][f] = S[b][t][f] * exp(2 pi i u[b][t][f])   (the random initial phases, u uniform in [0, 1))
__global__ void init_phase_kernel(const float *__restrict__ S, const float *__restrict__ u, v2f *__restrict__ spec, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float sn, cs;
        sincospif(2.0f * u[i], &sn, &cs);
        spec[i] = v2f{S[i] * cs, S[i] * sn};
    }
}

// one frame per workgroup: frames[b][t][n] = hann[n] * irfft(spec[b][t])[n]
template <int LOG2N>
__global__ __launch_bounds__(256) void istft_frames_kernel(const v2f *__restrict__ spec, float *__restrict__ frames, int F)
{
    constexpr int N = 1 << LOG2N;
    __shared__ v2f b0[N], b1[N], tw[N / 2];
    const int tid = threadIdx.x;
    const size_t fr = blockIdx.x;
    fill_twiddles<LOG2N>(tw, tid);
    const v2f *sp = spec + fr * F;
    for (int k = tid; k < N; k += 256) {
        v2f v;
        if (k <= N / 2) { v = sp[k]; if (k == 0 || k == N / 2) v.y = 0.f; }     // irfft ignores the imaginary part of DC / Nyquist
        else { v = sp[N - k]; v.y = -v.y; }
        b0[k] = v;
    }
    v2f *r = fft_lds<LOG2N>(b0, b1, tw, tid, true);
    float *dst = frames + fr * N;
    for (int n = tid; n < N; n += 256) {
        const float w = 0.5f - 0.5f * cospif(2.0f * (float)n / (float)N);
        dst[n] = r[n].x * (1.0f / (float)N) * w;
    }
}

// y[b][i] = (sum over frames covering sample i + N/2 of frames[b][t][i + N/2 - t hop]) / (sum of hann^2 over the same frames)
__global__ __launch_bounds__(256) void overlap_add_kernel(const float *__restrict__ frames, float *__restrict__ y, int B, int T, int N, int hop, int L)
{
    const int64_t total = (int64_t)B * L;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / L);
        const int n = (int)(i - (int64_t)b * L) + N / 2;
        int t1 = n / hop;
        if (t1 > T - 1) t1 = T - 1;
        int t0 = (n - N + hop) / hop;          // ceil((n - N + 1) / hop) for n - N + 1 > 0
        if (n - N + 1 <= 0) t0 = 0;
        float acc = 0.f, wss = 0.f;
        for (int t = t0; t <= t1; ++t) {
            const int k = n - t * hop;
            const float w = 0.5f - 0.5f * cospif(2.0f * (float)k / (float)N);
            acc += frames[((size_t)b * T + t) * N + k];
            wss += w * w;
        }
        y[i] = wss > 1.17549435e-38f ? acc / wss : acc;
    }
}

// one frame per workgroup: X = rfft(hann * reflect-padded frame of y);  spec[b][t][f] = S[b][t][f] * X[f] / |X[f]|
template <int LOG2N>
__global__ __launch_bounds__(256) void stft_phase_kernel(const float *__restrict__ y, const float *__restrict__ S, v2f *__restrict__ spec,
                                                         int T, int hop, int L, int F, int want_raw)
{
    constexpr int N = 1 << LOG2N;
    __shared__ v2f b0[N], b1[N], tw[N / 2];
    const int tid = threadIdx.x;
    const size_t fr = blockIdx.x;
    const int b = (int)(fr / T), t = (int)(fr - (size_t)b * T);
    fill_twiddles<LOG2N>(tw, tid);
    const float *yb = y + (size_t)b * L;
    for (int n = tid; n < N; n += 256) {
        int idx = t * hop + n - N / 2;
        if (idx < 0) idx = -idx;                              // np.pad(mode="reflect")
        if (idx >= L) idx = 2 * (L - 1) - idx;
        idx = idx < 0 ? 0 : (idx >= L ? L - 1 : idx);         // (signals shorter than N/2: clamp)
        const float w = 0.5f - 0.5f * cospif(2.0f * (float)n / (float)N);
        b0[n] = v2f{yb[idx] * w, 0.f};
    }
    v2f *r = fft_lds<LOG2N>(b0, b1, tw, tid, false);
    for (int f = tid; f < F; f += 256) {
        const v2f X = r[f];
        if (want_raw) { spec[fr * F + f] = X; continue; }
        const float mag = sqrtf(X.x * X.x + X.y * X.y);
        const float s = S[fr * F + f];
        spec[fr * F + f] = mag > 0.f ? v2f{s * X.x / mag, s * X.y / mag} : v2f{s, 0.f};   // np.angle(0) = 0
    }
}

// y[n] = x[n] + k y[n-1] is a decaying recurrence: the influence of y[n-W] on y[n] is k^W.  Each thread owns a chunk of
// CH samples and starts the recurrence W samples earlier from zero, W chosen so that k^W < 1e-9 (605 samples for
// k = 0.97): the discarded warm-up makes the chunks independent to far below fp32 rounding.  out-of-place.
constexpr int PRE_CHUNK = 2048;
__global__ void inv_preemphasis_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int L, float k, int warm)
{
    const int chunks = (L + PRE_CHUNK - 1) / PRE_CHUNK;
    const int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (id >= (int64_t)B * chunks) return;
    const int b = (int)(id / chunks), c = (int)(id - (int64_t)b * chunks);
    const float *px = x + (size_t)b * L;
    float *py = y + (size_t)b * L;
    const int n0 = c * PRE_CHUNK, n1 = min(L, n0 + PRE_CHUNK);
    float acc = 0.f;
    for (int n = max(0, n0 - warm); n < n0; ++n) acc = fmaf(k, acc, px[n]);
    for (int n = n0; n < n1; ++n) {
        acc = fmaf(k, acc, px[n]);
        py[n] = acc;
    }
}

inline int ew_blocks(int64_t n) { const int64_t b = nsg_cdiv(n, 256); return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }
inline int log2_of(int n) { int l = 0; while ((1 << l) < n) ++l; return (1 << l) == n ? l : -1; }

}  // namespace

extern "C" {

int nsg_audio_mel_to_linear(const float *mel, const float *inv_basis, float *S, int32_t B, int32_t n_mels, int32_t T, int32_t F,
                            float min_level_db, float ref_level_db, float max_abs_value, float power, void *stream)
{
    NSG_REQUIRE(mel && inv_basis && S && B > 0 && n_mels > 0 && T > 0 && F > 0 && max_abs_value > 0.f, NSG_E_INVALID, "nsg_audio_mel_to_linear: bad argument");
    NSG_REQUIRE((int64_t)B * T < 0x7fffffff, NSG_E_UNSUPPORTED, "nsg_audio_mel_to_linear: too many frames");
    hipLaunchKernelGGL(mel_to_linear_kernel, dim3((unsigned)(B * T)), dim3(256), (size_t)n_mels * sizeof(float), (hipStream_t)stream, mel, inv_basis, S,
                       B, n_mels, T, F, min_level_db, ref_level_db, max_abs_value, power);
    return nsg_check_launch("mel_to_linear_kernel");
}

size_t nsg_audio_griffin_lim_workspace_bytes(int32_t B, int32_t T, int32_t n_fft)
{
    if (B <= 0 || T <= 0 || n_fft <= 0) return 0;
    const size_t F = (size_t)n_fft / 2 + 1;
    return nsg_align_up((size_t)B * T * F * 2 * sizeof(float), 256) + nsg_align_up((size_t)B * T * n_fft * sizeof(float), 256);
}

// S [B][T][F] magnitudes, u [B][T][F] uniform numbers for the initial phases -> y [B][hop*(T-1)]
int nsg_audio_griffin_lim(const float *S, const float *u, float *y, int32_t B, int32_t T, int32_t n_fft, int32_t hop, int32_t iters,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    NSG_REQUIRE(S && u && y && B > 0 && T > 1 && hop > 0 && iters >= 0, NSG_E_INVALID, "nsg_audio_griffin_lim: bad argument");
    const int lg = log2_of(n_fft);
    NSG_REQUIRE(lg >= 9 && lg <= 11, NSG_E_UNSUPPORTED, "nsg_audio_griffin_lim: n_fft must be 512, 1024 or 2048");
    NSG_REQUIRE(n_fft % hop == 0, NSG_E_UNSUPPORTED, "nsg_audio_griffin_lim: hop must divide n_fft");
    NSG_REQUIRE(workspace && workspace_bytes >= nsg_audio_griffin_lim_workspace_bytes(B, T, n_fft), NSG_E_WORKSPACE, "nsg_audio_griffin_lim: workspace too small");
    NSG_REQUIRE((int64_t)B * T < 0x7fffffff, NSG_E_UNSUPPORTED, "nsg_audio_griffin_lim: too many frames");
    hipStream_t s = (hipStream_t)stream;
    const int F = n_fft / 2 + 1;
    const int L = hop * (T - 1);
    v2f *spec = reinterpret_cast<v2f *>(workspace);
    float *frames = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + nsg_align_up((size_t)B * T * F * 2 * sizeof(float), 256));
    const int64_t nspec = (int64_t)B * T * F;
    const unsigned nfr = (unsigned)(B * T);
    hipLaunchKernelGGL(init_phase_kernel, dim3(ew_blocks(nspec)), dim3(256), 0, s, S, u, spec, nspec);
    auto istft = [&]() {
        if (lg == 9)       hipLaunchKernelGGL((istft_frames_kernel<9>), dim3(nfr), dim3(256), 0, s, spec, frames, F);
        else if (lg == 10) hipLaunchKernelGGL((istft_frames_kernel<10>), dim3(nfr), dim3(256), 0, s, spec, frames, F);
        else               hipLaunchKernelGGL((istft_frames_kernel<11>), dim3(nfr), dim3(256), 0, s, spec, frames, F);
        hipLaunchKernelGGL(overlap_add_kernel, dim3(ew_blocks((int64_t)B * L)), dim3(256), 0, s, frames, y, B, T, n_fft, hop, L);
    };
    istft();
    for (int it = 0; it < iters; ++it) {
        if (lg == 9)       hipLaunchKernelGGL((stft_phase_kernel<9>), dim3(nfr), dim3(256), 0, s, y, S, spec, T, hop, L, F, 0);
        else if (lg == 10) hipLaunchKernelGGL((stft_phase_kernel<10>), dim3(nfr), dim3(256), 0, s, y, S, spec, T, hop, L, F, 0);
        else               hipLaunchKernelGGL((stft_phase_kernel<11>), dim3(nfr), dim3(256), 0, s, y, S, spec, T, hop, L, F, 0);
        istft();
    }
    return nsg_check_launch("griffin_lim");
}

// X [B][T][F] complex (interleaved re, im) = stft(y [B][L]) with T = 1 + L / hop frames (librosa.stft, centred, reflect, Hann)
int nsg_audio_stft(const float *y, float *X, int32_t B, int32_t L, int32_t n_fft, int32_t hop, void *stream)
{
    NSG_REQUIRE(y && X && B > 0 && L > 0 && hop > 0, NSG_E_INVALID, "nsg_audio_stft: bad argument");
    const int lg = log2_of(n_fft);
    NSG_REQUIRE(lg >= 9 && lg <= 11, NSG_E_UNSUPPORTED, "nsg_audio_stft: n_fft must be 512, 1024 or 2048");
    NSG_REQUIRE(L > n_fft / 2, NSG_E_UNSUPPORTED, "nsg_audio_stft: reflect padding needs more than n_fft/2 samples");
    const int T = 1 + L / hop, F = n_fft / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    const unsigned nfr = (unsigned)(B * T);
    v2f *spec = reinterpret_cast<v2f *>(X);
    if (lg == 9)       hipLaunchKernelGGL((stft_phase_kernel<9>), dim3(nfr), dim3(256), 0, s, y, nullptr, spec, T, hop, L, F, 1);
    else if (lg == 10) hipLaunchKernelGGL((stft_phase_kernel<10>), dim3(nfr), dim3(256), 0, s, y, nullptr, spec, T, hop, L, F, 1);
    else               hipLaunchKernelGGL((stft_phase_kernel<11>), dim3(nfr), dim3(256), 0, s, y, nullptr, spec, T, hop, L, F, 1);
    return nsg_check_launch("stft");
}

int nsg_audio_inv_preemphasis(const float *x, float *y, int32_t B, int32_t L, float k, void *stream)
{
    NSG_REQUIRE(x && y && x != y && B > 0 && L > 0, NSG_E_INVALID, "nsg_audio_inv_preemphasis: bad argument (out of place)");
    NSG_REQUIRE(fabsf(k) < 1.f, NSG_E_UNSUPPORTED, "nsg_audio_inv_preemphasis: |k| must be below 1 (a decaying filter)");
    int warm = 0;
    if (k != 0.f) {
        const double w = ceil(log(1e-9) / log(fabs((double)k)));
        warm = w > (double)L ? L : (int)w;
    }
    const int64_t threads = (int64_t)B * nsg_cdiv(L, PRE_CHUNK);
    hipLaunchKernelGGL(inv_preemphasis_kernel, dim3((unsigned)nsg_cdiv(threads, 64)), dim3(64), 0, (hipStream_t)stream, x, y, B, L, k, warm);
    return nsg_check_launch("inv_preemphasis_kernel");
}

}  // extern "C"
