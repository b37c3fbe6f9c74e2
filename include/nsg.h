/*
 * nsg.h -- C ABI of libnsg.so: hand-written gfx950 (MI355X / CDNA4) kernels for the VQ-VAE
 * training hot path of dendisuhubdy/neural_sound_generation.
 *
 * The reference has no native code on this path: it reaches its arithmetic through PyTorch ATen
 * operators.  Each entry point below therefore replaces one ATen call site of the reference
 * (file:line given per function, relative to the reference repository root).  The reference-side
 * binding a maintainer would add (ctypes, from src/models.py / src/vector_quantization.py) is
 * shown in INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (including
 *     workspaces); the library allocates nothing, frees nothing, keeps no pointer after return;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls only ENQUEUE work,
 *     they never synchronise;
 *   - return value: 0 = OK, <0 = invalid argument / unsupported shape (NSG_E_*), >0 = hipError_t;
 *     a message for the calling thread's last failure is available from nsg_last_error_string();
 *   - activations are fp32 NHWC ("channels last"): a tensor the reference holds as (B,C,H,W) is
 *     stored here as [B][H][W][C].  For C == 1 (the mel input and the reconstruction) the two
 *     layouts coincide.  Convolution weights are passed in the reference's own layouts
 *     (Conv2d: [C_out][C_in][kH][kW], ConvTranspose2d: [C_in][C_out][kH][kW]) and re-packed on the
 *     device by nsg_pack_conv_weights();
 *   - element counts of any single tensor must stay below 2^31.
 */
#ifndef NSG_H_
#define NSG_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define NSG_API __attribute__((visibility("default")))
#else
#define NSG_API
#endif

#define NSG_VERSION 102 /* bumped on ANY change of an existing entry point's signature; _lib.py refuses a library of another version */

enum {
    NSG_OK = 0,
    NSG_E_INVALID = -1,     /* null pointer / non-positive size / misaligned pointer */
    NSG_E_UNSUPPORTED = -2, /* shape outside what the kernels implement (see each function) */
    NSG_E_WORKSPACE = -3    /* workspace too small */
};

/* storage type of activations / packed weights: fp32 is the parity mode, bf16 (fp32 accumulate,
 * fp32 BatchNorm statistics, fp32 VQ) the throughput mode */
enum { NSG_F32 = 0, NSG_BF16 = 1 };

/* flags shared by the convolution entry points */
enum {
    NSG_RELU_IN = 1,   /* apply max(0,.) to the (gathered) input operand while loading it          */
    NSG_TANH_OUT = 2,  /* apply tanh to the result (forward only)                                   */
    NSG_RELU_IN2 = 4,  /* wgrad only: apply max(0,.) to the output-side operand (see nsg_conv_wgrad) */
    NSG_OUT_F32 = 8,   /* forward only: write y as fp32 even when the layer's dtype is bf16          */
    NSG_RELU_OUT = 16  /* forward only: apply max(0,.) to the result (the consumer's leading ReLU)   */
};

/* Geometry of one convolution layer.  transposed = 0: nn.Conv2d(C_in, C_out, k, stride, pad);
 * transposed = 1: nn.ConvTranspose2d(C_in, C_out, k, stride, pad).  (IH,IW) is the layer's input
 * extent, (OH,OW) its output extent; both are given so the library never has to guess.
 * Supported: Conv2d with any k <= 7 (rectangular: see k_w), stride in {1,2}; ConvTranspose2d with k=4, stride=2, pad=1
 * (the only transposed geometry on the path: src/models.py:179,182).  C_in and C_out must be
 * multiples of 4 or equal to 1. */
typedef struct nsg_conv_desc {
    int32_t B, IH, IW, C_in;
    int32_t OH, OW, C_out;
    int32_t k, stride, pad;
    int32_t transposed;
    int32_t dtype; /* NSG_F32 / NSG_BF16: storage type of the multi-channel activations and of the packed
                      weights.  Single-channel tensors (C == 1: the mel image, the reconstruction and their
                      gradients), biases, weight gradients and the reference-layout weights are always fp32.
                      bf16 needs channel counts that are multiples of 8. */
    int32_t k_w;   /* 0: square kernel (k x k, pad on both dimensions).  > 0: rectangular Conv2d (stride 1,
                      C_in > 1): k x k_w taps, padding (pad, pad_w), reference weight layout (C_out, C_in, k, k_w).
                      A stride-1 output may also be CROPPED at the bottom / right: OH / OW smaller than the full
                      extent (the pad-then-crop of the prior's masked stacks, src/models.py:268-273). */
    int32_t pad_w; /* column padding when k_w > 0 */
} nsg_conv_desc;

NSG_API int nsg_version(void);
NSG_API const char *nsg_last_error_string(void);

/* ---------------------------------------------------------------------------------------------
 * Vector quantiser                                        src/vector_quantization.py
 * ------------------------------------------------------------------------------------------- */

/* Bytes of workspace nsg_vq_forward needs for N rows and K codes. */
NSG_API size_t nsg_vq_workspace_bytes(int64_t N, int32_t D, int32_t K);

/* Fused nearest-code search.  Replaces VectorQuantization.forward (vector_quantization.py:6-23:
 * torch.sum(x**2), torch.sum(e**2), torch.addmm(..., alpha=-2, beta=1), torch.min(dim=1)) and the
 * gather of VectorQuantizationStraightThrough.forward (:40-42, torch.index_select), without ever
 * materialising the (N,K) distance matrix.
 *   x [N][D] fp32 rows, e [K][D] fp32 codebook, idx_out [N] int64 (first minimal index on ties,
 *   bit-exact with the reference's CPU result -- see DESIGN.md "bit-exact argmin"),
 *   codes_out [N][D] = e[idx] or NULL, dmin_out [N] fp32 minimal distance or NULL.
 * Supported: 1 <= D <= 256, K >= 1. */
NSG_API int nsg_vq_forward(const float *x, const float *e, int64_t N, int32_t D, int32_t K, int64_t *idx_out,
                           float *codes_out, float *dmin_out, void *workspace, size_t workspace_bytes, void *stream);

/* The same search for the bf16 compute mode: the contraction runs on the bf16 matrix pipe with both fp32 operands
 * split into bf16 hi + lo parts (3 MFMAs per 16 channels, fp32 accumulate): distances carry a relative error of
 * ~2^-16, so on near-ties the chosen code may differ from the reference's (NOT the bit-exact search; the fp32
 * parity mode uses nsg_vq_forward).  D % 8 == 0.  Workspace: nsg_vq_bf16x3_workspace_bytes.
 * codes_bf16_out (or NULL): the chosen code rows as bf16 [N][D], with max(0, .) applied when bf16_relu != 0 -- the
 * decoder's input after its leading ReLU (models.py:176) in the bf16 mode, written here instead of by a separate
 * conversion pass over an fp32 copy (codes_out may then be NULL: nsg_vq_losses_indexed reads the codebook itself). */
NSG_API size_t nsg_vq_bf16x3_workspace_bytes(int64_t N, int32_t D, int32_t K);
NSG_API int nsg_vq_forward_bf16x3(const float *x, const float *e, int64_t N, int32_t D, int32_t K, int64_t *idx_out,
                                  float *codes_out, float *dmin_out, void *codes_bf16_out, int32_t bf16_relu, void *workspace,
                                  size_t workspace_bytes, void *stream);
/* The same with a per-clip conditioning row folded into the bf16 code write (BASELINE configs[2], the speaker-conditioned
 * decoder; an extension: the reference loads the speaker id and ignores it, src/train.py:114): row n belongs to clip
 * n / rows_per_clip and codes_bf16_out[n] = [relu]( e[idx[n]] + clip_rows[clip] ), clip_rows fp32 [clips][D].  idx_out,
 * codes_out and dmin_out are unaffected.  clip_rows == NULL is nsg_vq_forward_bf16x3. */
NSG_API int nsg_vq_forward_bf16x3_cond(const float *x, const float *e, int64_t N, int32_t D, int32_t K, int64_t *idx_out,
                                       float *codes_out, float *dmin_out, void *codes_bf16_out, int32_t bf16_relu,
                                       const float *clip_rows, int64_t rows_per_clip, void *workspace, size_t workspace_bytes,
                                       void *stream);

/* out[r] = torch.sum(v[r]**2) with ATen's CPU summation order (vector_quantization.py:12-13). */
NSG_API int nsg_rowsumsq(const float *v, int64_t rows, int32_t D, float *out, void *stream);

/* Bytes of workspace for nsg_index_add_rows. */
NSG_API size_t nsg_index_add_workspace_bytes(int64_t N, int32_t D, int32_t K);

/* out[k][:] = sum over rows i with idx[i] == k of g[i][:]  (out fully overwritten).
 * Replaces grad_codebook.index_add_(0, indices, grad_output) (vector_quantization.py:60-61) and the
 * autograd of torch.index_select(self.embedding.weight, 0, indices) (src/models.py:137-138); also
 * yields the per-code sums of the EMA codebook extension.  Deterministic (fixed summation order).
 * counts_out [K] fp32 (rows per code) or NULL. */
NSG_API int nsg_index_add_rows(const int64_t *idx, const float *g, int64_t N, int32_t D, int32_t K, float *out,
                               float *counts_out, void *workspace, size_t workspace_bytes, void *stream);

/* The same for the bf16 compute mode: the one-hot reduction runs on the bf16 matrix pipe with g split into bf16
 * hi + lo parts (relative error of a sum ~2^-17; deterministic).  Falls back to nsg_index_add_rows's kernel when
 * D is not a multiple of 8 or D <= 32.  Same workspace. */
NSG_API int nsg_index_add_rows_bf16x2(const int64_t *idx, const float *g, int64_t N, int32_t D, int32_t K, float *out,
                                      float *counts_out, void *workspace, size_t workspace_bytes, void *stream);

/* The same sums as a sorted segment sum: a stable counting sort of the row INDICES by code, then every code's rows added in
 * row order (fp32, bitwise reproducible).  Moves N*D*4 bytes instead of doing 2*N*K*D flops: the form for large codebooks
 * (K = 8192) and the default of the training step.  Rows whose index lies outside [0, K) contribute nothing.
 * Needs D % 4 == 0 with D / 4 a power of two up to 64 (or D a multiple of 256), K <= 8192, N < 2^31. */
NSG_API size_t nsg_index_add_sorted_workspace_bytes(int64_t N, int32_t D, int32_t K);
NSG_API int nsg_index_add_rows_sorted(const int64_t *idx, const float *g, int64_t N, int32_t D, int32_t K, float *out,
                                      float *counts_out, void *workspace, size_t workspace_bytes, void *stream);

/* out[i][:] = e[idx[i]][:] for i < N.  Replaces torch.index_select(codebook, 0, indices)
 * (vector_quantization.py:40-41, src/models.py:137) and self.codebook.embedding(latents)
 * (src/models.py:194).  Indices outside [0,K) are clamped. */
NSG_API int nsg_gather_rows(const float *e, const int64_t *idx, int64_t N, int32_t D, int32_t K, float *out,
                            void *stream);

/* Codebook gradient of loss_vq = mse(codebook[idx], sg(z_e)) (src/train.py:131 through index_select, src/models.py:137)
 * from per-code statistics instead of an (N, D) gradient tensor: with n[k] rows assigned to code k and s[k] = their sum
 * (nsg_index_add_rows over z_e with counts),  out[k][:] = scale * (n[k] * e[k][:] - s[k][:]),  scale = 2 / (N*D). */
NSG_API int nsg_codebook_grad_from_sums(const float *e, const float *n, const float *s, int32_t K, int32_t D, float scale,
                                        float *out, void *stream);

/* EMA codebook update (extension, not in the reference; VQ-VAE paper appendix A.1):
 *   ema_n = decay*ema_n + (1-decay)*n;  ema_s = decay*ema_s + (1-decay)*s;
 *   e[k] = ema_s[k] / ((ema_n[k]+eps) / (sum(ema_n) + K*eps) * sum(ema_n)).
 * n [K], s [K][D] are the (all-reduced) batch statistics from nsg_index_add_rows.
 * scratch: one device float (holds sum(ema_n) between the two kernels). */
NSG_API int nsg_vq_ema_update(float *e, float *ema_n, float *ema_s, const float *n, const float *s, int32_t K,
                              int32_t D, float decay, float eps, float *scratch, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Convolutions                                            src/models.py:150,153,165,168,179,182
 * ------------------------------------------------------------------------------------------- */

/* ELEMENTS (of d->dtype) to allocate for each packed weight image (forward image, dgrad image).
 * (The two single-channel layers keep one of their images as fp32 [C][16] for the stencil kernels whatever
 * d->dtype is; bf16 layers whose channel counts the patch-staged kernel takes -- a multiple of 128 on one side and of
 * 64 on the other -- carry a second, fragment-ordered copy behind the plain [tap][n][c] image, which the bf16 conv
 * forward / data gradient reads: the count accounts for both.  Packed images are only ever produced by
 * nsg_pack_conv_weights(_batch) into buffers of exactly this size.) */
NSG_API size_t nsg_packed_weight_floats(const nsg_conv_desc *d);

/* Re-pack reference-layout weights into the two [tap][n][c] images the GEMM kernels stream:
 * w_fwd for nsg_conv_forward, w_dgrad for nsg_conv_dgrad (either may be NULL to skip it). */
NSG_API int nsg_pack_conv_weights(const nsg_conv_desc *d, const float *w, void *w_fwd, void *w_dgrad, void *stream);

/* The same for n layers in ONE kernel launch (a training step re-packs every layer after the optimiser
 * step): descs[n], w[n], w_fwd[n], w_dgrad[n] are host arrays; NULL image pointers are skipped. */
NSG_API int nsg_pack_conv_weights_batch(int32_t n, const nsg_conv_desc *descs, const float *const *w,
                                        void *const *w_fwd, void *const *w_dgrad, void *stream);

/* Workspace bytes for forward/dgrad (C == 1 layers stage im2col/col2im images) and wgrad (split-K slabs). */
NSG_API size_t nsg_conv_workspace_bytes(const nsg_conv_desc *d);

/* y = conv(x) + bias.  Replaces F.conv2d / F.conv_transpose2d as called by nn.Conv2d /
 * nn.ConvTranspose2d.forward at src/models.py:150,153,165,168,179,182.
 * flags: NSG_RELU_IN (the preceding nn.ReLU fused into the load), NSG_TANH_OUT (models.py:183).
 * CONTRACT for w_fwd here and w_dgrad in nsg_conv_dgrad*: the pointer is an image written by nsg_pack_conv_weights(_batch) for
 * the SAME descriptor into a buffer of nsg_packed_weight_floats(d) elements.  For the bf16 shapes the patch-staged kernel takes
 * that count is TWICE taps * C_out * C_in: the kernel reads the fragment-ordered copy that the packer put behind the plain image
 * and cannot tell a shorter caller-made buffer from a packed one (it would read past it).  tests/test_abi.py pins the sizes. */
NSG_API int nsg_conv_forward(const nsg_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y,
                             int32_t flags, void *workspace, size_t workspace_bytes, void *stream);

/* nsg_conv_forward plus the training-mode BatchNorm statistics of its output y in the same pass:
 * nn.Conv2d / nn.ConvTranspose2d followed by nn.BatchNorm2d (src/models.py:165-166, 150-151,
 * 153-154, 179-180).  The conv epilogue reduces each 128-row output tile to (count, mean, M2) while
 * the tile is still in LDS; the tiles are merged exactly as nsg_bn_stats merges its slabs.
 * mean/invstd/running_* as in nsg_bn_stats.  NSG_TANH_OUT is not allowed here. */
NSG_API int nsg_conv_forward_bnstats(const nsg_conv_desc *d, const void *x, const void *w_fwd, const float *bias,
                                     void *y, int32_t flags, float eps, float momentum, float *mean, float *invstd,
                                     float *running_mean, float *running_var, void *workspace, size_t workspace_bytes,
                                     void *stream);

/* dx = d loss / d x given dy (autograd of the calls above).  flags: none. */
NSG_API int nsg_conv_dgrad(const nsg_conv_desc *d, const void *dy, const void *w_dgrad, void *dx, int32_t flags,
                           void *workspace, size_t workspace_bytes, void *stream);

/* The same with the rest of a ResBlock's input gradient folded into the kernel's store:
 *   dx = (conv_dgrad(dy) + add) * (relu_x > 0)
 * add (or NULL): the gradient arriving over the skip connection; relu_x (or NULL): the ReLU'd tensor the
 * convolution read (src/models.py:149,158: x is overwritten by relu(x), so mask = relu_x > 0).  Both are laid
 * out like dx.  Replaces nsg_conv_dgrad + nsg_relu_backward_add (one write and one read of dx less). */
NSG_API int nsg_conv_dgrad_relu_add(const nsg_conv_desc *d, const void *dy, const void *w_dgrad, const void *add,
                                    const void *relu_x, void *dx, int32_t flags, void *workspace,
                                    size_t workspace_bytes, void *stream);

/* dw (reference weight layout, fully overwritten) and dbias (or NULL) given x and dy.
 * flags: NSG_RELU_IN treats x as max(0,x) (the fused preceding ReLU).  Deterministic. */
NSG_API int nsg_conv_wgrad(const nsg_conv_desc *d, const void *x, const void *dy, float *dw, float *dbias,
                           int32_t flags, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d over [M][C] (M = B*H*W)                     src/models.py:151,154,166,180
 * ------------------------------------------------------------------------------------------- */

NSG_API size_t nsg_bn_workspace_bytes(int64_t M, int32_t C);

/* Training-mode statistics: mean[C], invstd[C] = 1/sqrt(biased var + eps); if running_mean/var are
 * non-NULL they are updated with `momentum` (running_var takes the unbiased variance), as
 * F.batch_norm(training=True) does. */
NSG_API int nsg_bn_stats(const void *x, int64_t M, int32_t C, int32_t dtype, float eps, float momentum, float *mean,
                         float *invstd, float *running_mean, float *running_var, void *workspace,
                         size_t workspace_bytes, void *stream);

/* Eval mode: mean = running_mean, invstd = 1/sqrt(running_var + eps). */
NSG_API int nsg_bn_eval_stats(const float *running_mean, const float *running_var, int32_t C, float eps, float *mean,
                              float *invstd, void *stream);

/* (BatchNorm entry points: x / residual / y_relu / dy / dx hold elements of `dtype` (NSG_F32 or NSG_BF16,
 * void* below; nsg_bn_apply may write y in a different y_dtype); statistics, per-channel parameters and all
 * arithmetic are fp32.  bf16 needs C % 8 == 0.) */
/* y = (x-mean)*invstd*gamma + beta; relu & 1: y = max(0,y); residual != NULL: y += residual
 * (relu_residual != 0: y += max(0,residual) -- the ResBlock's in-place-ReLU'd skip, models.py:149,158);
 * relu & 2: y = max(0,y) once more at the very end -- the NEXT ResBlock's leading in-place ReLU
 * (models.py:149) applied where the tensor is produced instead of where it is consumed. */
NSG_API int nsg_bn_apply(const void *x, const float *mean, const float *invstd, const float *gamma, const float *beta,
                         const void *residual, void *y, int64_t M, int32_t C, int32_t relu, int32_t relu_residual,
                         int32_t dtype, int32_t y_dtype, void *stream);

/* Backward of the call above with respect to x, gamma, beta.  When the forward used relu & 1, the ReLU mask
 * comes from ONE of: relu_beta = the forward's beta [C] (the mask (x-mean)*(invstd*gamma)+beta > 0 is then
 * re-derived from x with the forward's own arithmetic and y_relu is not read -- one tensor less of traffic),
 * or y_relu = the forward OUTPUT (its sign is the mask).  Both NULL: no ReLU.  dgamma/dbeta [C] overwritten.
 * dx_colsum [C] or NULL: column sums of dx, i.e. the bias gradient of the convolution that feeds
 * this BatchNorm (autograd of nn.Conv2d's bias at src/models.py:150,153,165,179), produced by the
 * kernel that writes dx instead of a second pass over it. */
NSG_API int nsg_bn_backward(const void *x, const void *y_relu, const void *dy, const float *mean,
                            const float *invstd, const float *gamma, const float *relu_beta, void *dx, float *dgamma,
                            float *dbeta, float *dx_colsum, int64_t M, int32_t C, int32_t dtype, void *workspace,
                            size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The single-channel input layer with its BatchNorm, as one operator     src/models.py:165-167
 *   encoder.0 Conv2d(1, C, 4, 2, 1) -> encoder.1 BatchNorm2d(C) -> encoder.2 ReLU(True)
 * The convolution's output is 2C times larger than the image it comes from and costs 16 multiply-adds
 * per element, so it is never stored: the statistics pass, the apply pass and both backward passes
 * recompute it from the image (one tensor write forward, two tensor reads backward, against three
 * writes and eight reads of nsg_conv_forward + nsg_bn_stats + nsg_bn_apply + nsg_bn_backward +
 * nsg_conv_wgrad, whose results these reproduce: same value for h bit for bit, same expressions).
 *   img [B][H][W] fp32 (H, W even); w the parameter's own layout (C,1,4,4) = [C][16] fp32; bias [C] or NULL;
 *   y / dy [B][H/2][W/2][C] of y_dtype / dy_dtype (NSG_F32 or NSG_BF16); C % 4 == 0, C <= 1024.
 * ------------------------------------------------------------------------------------------- */
NSG_API size_t nsg_c1conv_bn_workspace_bytes(int32_t C);
/* training != 0: mean / invstd [C] are OUTPUTS (batch statistics of the conv output; running_mean / running_var
 * updated with `momentum` when not NULL, unbiased variance as nn.BatchNorm2d); training == 0: they are INPUTS
 * (nsg_bn_eval_stats) and the workspace is not used.  y = max(0, (h - mean) * (invstd * gamma) + beta). */
NSG_API int nsg_c1conv_bn_relu_forward(const float *img, const float *w, const float *bias, const float *gamma, const float *beta,
                                       float *mean, float *invstd, float *running_mean, float *running_var, float eps,
                                       float momentum, int32_t training, void *y, int32_t y_dtype, int32_t B, int32_t H,
                                       int32_t W, int32_t C, void *workspace, size_t workspace_bytes, double *moments,
                                       void *stream);
/* Gradients of the layer's parameters from dy = dL/dy (the image is data: no input gradient):
 * dw [C][16], dbias [C] or NULL, dgamma [C], dbeta [C], all overwritten; mean / invstd as the forward left them.
 * moments (both calls; NULL allowed): NSG_C1_MOMENTS doubles = the image's tap moments S[t] = sum_p x[p][t] and
 * P[t][u] = sum_p x[p][t] x[p][u] over the 16 patch values of every output pixel (stored for values shifted by a constant,
 * the last entry: conditioning).  The bf16 layer takes its batch statistics
 * and its weight gradient from them (one pass over dy instead of two).  The training forward writes them when the pointer is
 * not NULL; the backward reads them when given and recomputes them from the image otherwise. */
NSG_API int nsg_c1conv_bn_relu_backward(const float *img, const float *w, const float *bias, const float *gamma,
                                        const float *beta, const float *mean, const float *invstd, const void *dy,
                                        int32_t dy_dtype, float *dw, float *dbias, float *dgamma, float *dbeta, int32_t B,
                                        int32_t H, int32_t W, int32_t C, void *workspace, size_t workspace_bytes,
                                        const double *moments, void *stream);
#define NSG_C1_MOMENTS 273

/* ---------------------------------------------------------------------------------------------
 * The single-channel OUTPUT layer with the BatchNorm in front of it, as one operator     src/models.py:180-183
 *   decoder.4 BatchNorm2d(C) -> decoder.5 ReLU(True) -> decoder.6 ConvTranspose2d(C, 1, 4, 2, 1) [-> decoder.7 Tanh]
 * u [B][H][W][C] is the BatchNorm INPUT (the transposed conv in front wrote it; its batch statistics come from
 * nsg_bn_stats).  Neither the activated tensor a = relu(bn(u)) nor the data gradient of the transposed conv is ever
 * stored: the forward applies BatchNorm + ReLU on the operand's way into the MFMA, the backward rebuilds the
 * data gradient from the gradient image (16 taps per element, on the matrix cores) inside the BatchNorm-backward
 * passes and the weight gradient from a rebuilt on the spot -- 4 tensor reads + 1 write against 8 reads + 3 writes
 * of nsg_bn_apply + nsg_conv_forward + nsg_conv_wgrad + nsg_conv_dgrad + nsg_bn_backward.
 * Available for bf16 tensors with C = 32, 64, 96, 128 (nsg_bn_relu_c1convt_supported); other shapes use the
 * separate operators.  w: the parameter's own layout (C,1,4,4) = [C][16] fp32; y / dy: [B][2H][2W] fp32.
 * ------------------------------------------------------------------------------------------- */
NSG_API int32_t nsg_bn_relu_c1convt_supported(int32_t dtype, int32_t C);
NSG_API size_t nsg_bn_relu_c1convt_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C);
/* y = [tanh](bias + convT(relu((u - mean) * invstd * gamma + beta)));  flags: 0 or NSG_TANH_OUT */
NSG_API int nsg_bn_relu_c1convt_forward(const void *u, int32_t dtype, const float *mean, const float *invstd, const float *gamma,
                                        const float *beta, const float *w, const float *bias, float *y, int32_t flags, int32_t B,
                                        int32_t H, int32_t W, int32_t C, void *workspace, size_t workspace_bytes, void *stream);
/* The same with the Tanh, plus the reconstruction loss of src/train.py:118-129 taken in the pass that writes the image:
 * target [B][2H][T] fp32 with T >= 2W (x_tilde is zero-padded on the right to the target's width, as the reference pads it on the
 * host); loss_out[0] = mean((pad(x_tilde) - target)^2); dpre [B][2H][2W] = grad_scale * 2 / (B 2H T) * (x_tilde - target) *
 * (1 - x_tilde^2) = the gradient w.r.t. the Tanh's INPUT (what nsg_bn_relu_c1convt_backward takes as dy); y = x_tilde is stored
 * only when not NULL; dbias [1] or NULL = sum of dpre, the transposed conv's bias gradient (pass dbias = NULL to the backward
 * then).  Replaces nsg_bn_relu_c1convt_forward + nsg_mse_padded + nsg_tanh_backward in a training step. */
NSG_API int nsg_bn_relu_c1convt_forward_mse(const void *u, int32_t dtype, const float *mean, const float *invstd, const float *gamma,
                                            const float *beta, const float *w, const float *bias, float *y, const float *target,
                                            int32_t T, float grad_scale, float *loss_out, float *dpre, float *dbias, int32_t B,
                                            int32_t H, int32_t W, int32_t C, void *workspace, size_t workspace_bytes, void *stream);
/* dy: gradient w.r.t. the transposed conv's output (BEFORE the tanh: nsg_tanh_backward).  Outputs: du (dtype, the
 * gradient w.r.t. u), du_colsum [C] or NULL (column sums of du = the bias gradient of the conv that wrote u, as
 * nsg_bn_backward's dx_colsum), dw [C][16], dbias [1] or NULL, dgamma [C], dbeta [C], all overwritten. */
NSG_API int nsg_bn_relu_c1convt_backward(const void *u, int32_t dtype, const float *mean, const float *invstd, const float *gamma,
                                         const float *beta, const float *w, const float *dy, void *du, float *du_colsum, float *dw,
                                         float *dbias, float *dgamma, float *dbeta, int32_t B, int32_t H, int32_t W, int32_t C,
                                         void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The ResBlock's 1x1 convolution with the BatchNorm work around it folded in     src/models.py:151-155
 *   ... BatchNorm2d(dim) -> ReLU(True) -> Conv2d(dim, dim, 1) -> BatchNorm2d(dim)
 * A 1x1 conv over NHWC rows is a flat GEMM [M][C] x [C][C]: a stream kernel.  These entry points apply the
 * BatchNorm arithmetic of the neighbouring layers on the operand's way from the global-load registers into LDS,
 * so the activated tensor relu(bn(x)) is never stored and the second BatchNorm's input gradient is produced and
 * consumed in one pass.  bf16 tensors, C = 32, 64, 128 (nsg_bn_relu_conv1x1_supported); w: (C, C, 1, 1) fp32.
 * ------------------------------------------------------------------------------------------- */
NSG_API int32_t nsg_bn_relu_conv1x1_supported(int32_t dtype, int32_t C);
NSG_API size_t nsg_bn_relu_conv1x1_workspace_bytes(int64_t M, int32_t C);
/* y = relu((x - mean) * invstd * gamma + beta) * w^T + bias          (replaces nsg_bn_apply + nsg_conv_forward) */
NSG_API int nsg_bn_relu_conv1x1_forward(const void *x, const float *mean, const float *invstd, const float *gamma, const float *beta,
                                        const float *w, const float *bias, void *y, int64_t M, int32_t C, int32_t dtype,
                                        void *workspace, size_t workspace_bytes, void *stream);
/* The same, plus the batch statistics of y taken from the store phase (what nsg_bn_stats(y) computes, for the BatchNorm
 * that follows): mean_y / invstd_y [C] out, running statistics updated with `momentum` when not NULL. */
NSG_API int nsg_bn_relu_conv1x1_forward_bnstats(const void *x, const float *mean, const float *invstd, const float *gamma,
                                                const float *beta, const float *w, const float *bias, void *y, float eps,
                                                float momentum, float *mean_y, float *invstd_y, float *running_mean_y,
                                                float *running_var_y, int64_t M, int32_t C, int32_t dtype, void *workspace,
                                                size_t workspace_bytes, void *stream);
/* dw[o][i] = sum_m dy[m][o] * relu(bn(x))[m][i]                        (nsg_conv_wgrad with the activation rebuilt from x) */
NSG_API int nsg_bn_relu_conv1x1_wgrad(const void *x, const float *mean, const float *invstd, const float *gamma, const float *beta,
                                      const void *dy, float *dw, int64_t M, int32_t C, int32_t dtype, void *workspace,
                                      size_t workspace_bytes, void *stream);
/* The apply half of nsg_bn_backward (no ReLU; dgamma / dbeta from nsg_bn_backward_sums) fused with the data gradient of
 * the 1x1 conv in FRONT of that BatchNorm:  dh = bn-backward(dy) at input h (stored: the weight gradient needs it),
 * dx = dh * w, dh_colsum [C] or NULL = column sums of dh (the conv's bias gradient).
 * prev_x != NULL: the conv's input was relu(bn_prev(prev_x)); the sums of THAT BatchNorm's backward over (prev_x, dx) --
 * what nsg_bn_backward_sums(prev_x, dx, ..., relu_beta = prev_beta) computes -- are formed while dx is written:
 * prev_dgamma / prev_dbeta [C] out (then nsg_bn_backward_apply finishes that BatchNorm without a reduction pass). */
NSG_API int nsg_bn_backward_conv1x1_dgrad(const void *h, const void *dy, const float *mean, const float *invstd, const float *gamma,
                                          const float *dgamma, const float *dbeta, const float *w, void *dh, void *dx,
                                          float *dh_colsum, const void *prev_x, const float *prev_mean, const float *prev_invstd,
                                          const float *prev_gamma, const float *prev_beta, float *prev_dgamma, float *prev_dbeta,
                                          int64_t M, int32_t C, int32_t dtype, void *workspace, size_t workspace_bytes, void *stream);
/* nsg_bn_backward_conv1x1_dgrad + nsg_bn_relu_conv1x1_wgrad in ONE pass over the tensors (bf16, C = 128; the BatchNorm in front is
 * required): dx, dh_colsum (or NULL), prev_dgamma / prev_dbeta as above, and dw[o][i] = sum_m dh[m][o] * relu(bn_prev(prev_x))[m][i]
 * (the autograd of src/models.py:153 for the conv weight).  dh is never stored.  4 tensor passes instead of 7. */
NSG_API int32_t nsg_bn_backward_conv1x1_dgrad_wgrad_supported(int32_t dtype, int32_t C);
NSG_API size_t nsg_bn_backward_conv1x1_dgrad_wgrad_workspace_bytes(int64_t M, int32_t C);
NSG_API int nsg_bn_backward_conv1x1_dgrad_wgrad(const void *h, const void *dy, const float *mean, const float *invstd, const float *gamma,
                                                const float *dgamma, const float *dbeta, const float *w, void *dx, float *dw,
                                                float *dh_colsum, const void *prev_x, const float *prev_mean, const float *prev_invstd,
                                                const float *prev_gamma, const float *prev_beta, float *prev_dgamma, float *prev_dbeta,
                                                int64_t M, int32_t C, int32_t dtype, void *workspace, size_t workspace_bytes,
                                                void *stream);
/* The apply half of nsg_bn_backward alone: dgamma / dbeta are inputs. */
NSG_API int nsg_bn_backward_apply(const void *x, const void *y_relu, const void *dy, const float *mean, const float *invstd,
                                  const float *gamma, const float *relu_beta, const float *dgamma, const float *dbeta, void *dx,
                                  float *dx_colsum, int64_t M, int32_t C, int32_t dtype, void *workspace, size_t workspace_bytes,
                                  void *stream);
/* The reduction half of nsg_bn_backward alone (same arguments, same values): dgamma, dbeta. */
NSG_API int nsg_bn_backward_sums(const void *x, const void *y_relu, const void *dy, const float *mean, const float *invstd,
                                 const float *gamma, const float *relu_beta, float *dgamma, float *dbeta, int64_t M, int32_t C,
                                 int32_t dtype, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Element-wise, losses, optimiser                         src/train.py:118-136
 * ------------------------------------------------------------------------------------------- */

/* dx = (a + b) * (x > 0)   (b may be NULL): gradient through the ResBlock's leading in-place ReLU.
 * All four tensors hold elements of `dtype`. */
NSG_API int nsg_relu_backward_add(const void *a, const void *b, const void *x, void *dx, int64_t n, int32_t dtype,
                                  void *stream);

/* dst = src (relu != 0: max(0, src)) with a change of storage type (fp32 <-> bf16, round to nearest even). */
NSG_API int nsg_convert(const void *src, int32_t src_dtype, void *dst, int32_t dst_dtype, int64_t n, int32_t relu,
                        void *stream);

/* dx = g * (1 - y*y): backward of nn.Tanh (models.py:183) from its output y. */
NSG_API int nsg_tanh_backward(const float *g, const float *y, float *dx, int64_t n, void *stream);

/* *counters[i] += 1 for i < n: the `num_batches_tracked += 1` of every nn.BatchNorm2d a training-mode forward passes
 * (src/models.py:151,154,166,180) as one launch per 32 counters.  `counters` is a HOST array of device pointers. */
NSG_API int nsg_increment_counters(int64_t *const *counters, int32_t n, void *stream);

/* y = a + b (b may be NULL -> copy). */
NSG_API int nsg_add(const float *a, const float *b, float *y, int64_t n, void *stream);

/* Speaker-conditioned decoder (extension; BASELINE configs[2]; not in the reference, whose VQVAE
 * ignores the speaker id g, src/train.py:114):  y[b][r][:] = x[b][r][:] + rows[b][:] for the
 * rows_per_clip pixels r of clip b, and its gradient w.r.t. rows: out[b][:] = sum_r x[b][r][:]. */
NSG_API int nsg_add_per_clip(const float *x, const float *rows, void *y, int32_t B, int64_t rows_per_clip, int32_t C,
                             int32_t y_dtype, void *stream);
NSG_API size_t nsg_clip_colsum_workspace_bytes(int32_t B, int32_t C);
NSG_API int nsg_clip_colsum(const void *x, int32_t dtype, int32_t B, int64_t rows_per_clip, int32_t C, float *out,
                            void *workspace, size_t workspace_bytes, void *stream);

NSG_API size_t nsg_reduce_workspace_bytes(int64_t n);

/* loss_out[0] = mean over rows*wc elements of (pad(a) - c)^2 where a is [rows][wa], c is [rows][wc],
 * wa <= wc and a is zero-padded on the right to wc: the reference's zero-pad + F.mse_loss
 * (src/train.py:118-129).  da (or NULL) receives grad_scale * 2/(rows*wc) * (a - c[:, :wa]). */
NSG_API int nsg_mse_padded(const float *a, const float *c, int64_t rows, int32_t wa, int32_t wc, float grad_scale,
                           float *loss_out, float *da, void *workspace, size_t workspace_bytes, void *stream);

/* loss_out[0] = mean((q - z)^2) over n elements (train.py:131,133: both terms have this value);
 * dz (or NULL) = dz_scale * 2/n * (z - q) (+ dz_add if non-NULL: the straight-through gradient),
 * dq (or NULL) = dq_scale * 2/n * (q - z).  z, q, dq are fp32 (the quantiser works in fp32 in both
 * modes); dz and dz_add hold elements of grad_dtype (the encoder-side activation gradient). */
NSG_API int nsg_vq_losses(const float *z, const float *q, int64_t n, float dz_scale, float dq_scale,
                          const void *dz_add, float *loss_out, void *dz, float *dq, int32_t grad_dtype,
                          void *workspace, size_t workspace_bytes, void *stream);

/* The same loss and encoder-side gradient with q given by (codebook [K][D], idx [N]): q[r] = codebook[idx[r]] is read from
 * the cache-resident codebook instead of a materialised tensor (D % 8 == 0).  The codebook-side gradient of the loss is
 * 2/n * (n_k e_k - sum of the rows assigned to k): nsg_index_add_rows over z with counts (train.py:131). */
NSG_API int nsg_vq_losses_indexed(const float *z, const float *codebook, const int64_t *idx, int64_t N, int32_t D, int32_t K,
                                  float dz_scale, const void *dz_add, float *loss_out, void *dz, int32_t grad_dtype,
                                  void *workspace, size_t workspace_bytes, void *stream);
/* The same when dz is the incoming gradient of a BatchNorm whose input is bn_x [N][D] of grad_dtype (the encoder's last
 * ResBlock ends in one, src/models.py:154): that BatchNorm's backward sums over the dz values as stored -- bn_dbeta[d] = sum_n
 * dz[n][d], bn_dgamma[d] = sum_n dz[n][d] * (bn_x[n][d] - bn_mean[d]) * bn_invstd[d], what nsg_bn_backward_sums(bn_x, dz)
 * returns, bit for bit when grad_dtype is NSG_BF16 (the same slabs, the same order) -- are formed while dz is written, instead of
 * a pass that reads dz and bn_x back.  D: a multiple of 8 up to 2048 (nsg_vq_losses_indexed_bn_supported); dz must not be NULL. */
NSG_API int32_t nsg_vq_losses_indexed_bn_supported(int32_t D);
NSG_API size_t nsg_vq_losses_indexed_bn_workspace_bytes(int64_t N, int32_t D);
NSG_API int nsg_vq_losses_indexed_bn(const float *z, const float *codebook, const int64_t *idx, int64_t N, int32_t D, int32_t K,
                                     float dz_scale, const void *dz_add, float *loss_out, void *dz, int32_t grad_dtype,
                                     const void *bn_x, const float *bn_mean, const float *bn_invstd, float *bn_dgamma,
                                     float *bn_dbeta, void *workspace, size_t workspace_bytes, void *stream);

/* torch.optim.Adam step (src/main.py:124 defaults, no weight decay, no amsgrad) over one flat
 * fp32 buffer.  g is multiplied by grad_scale first (1/world_size after a sum all-reduce).
 * step is the 1-based step count. */
NSG_API int nsg_adam_step(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2,
                          float eps, int32_t step, float grad_scale, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Self-checks used by the tests
 * ------------------------------------------------------------------------------------------- */

/* out[i*K+k] = x[i]·e[k] evaluated (mode 0) as a sequential fmaf chain on the vector ALU and
 * (mode 1) on v_mfma_f32_32x32x2_f32 in the order nsg_vq_forward uses; the two must agree bit for
 * bit for the argmin to be exact.  N, K multiples of 32, D <= 256. */
NSG_API int nsg_debug_dot(const float *x, const float *e, int32_t N, int32_t D, int32_t K, int32_t mode, float *out,
                          void *stream);

/* nsg_vq_forward with the contraction on the vector ALU (an explicit fmaf chain per (row, code)) instead of the matrix
 * cores: same arguments, same results bit for bit -- the cross-check that the MFMA form keeps the reference's
 * summation order (src/vector_quantization.py:12-19).  Slow; tests only. */
NSG_API int nsg_debug_vq_forward_valu(const float *x, const float *e, int64_t N, int32_t D, int32_t K, int64_t *idx_out,
                                      float *codes_out, float *dmin_out, void *workspace, size_t workspace_bytes,
                                      void *stream);

/* ---------------------------------------------------------------------------------------------
 * Latent prior (GatedPixelCNN over the code-index grid)              src/models.py:219-341
 * Its convolutions are nsg_conv_* calls (rectangular masked kernels embedded in square ones);
 * these are the element-wise pieces.  fp32, rows [M][channels].
 * ------------------------------------------------------------------------------------------- */

/* GatedActivation with the class-conditional add folded in (models.py:219-226,268,274):
 *   y[m][c] = tanh(x[m][c] + cond[b][c]) * sigmoid(x[m][C+c] + cond[b][C+c]),  b = m / rows_per_clip.
 * x [M][2C], cond [B][2C] or NULL, y [M][C].  C % 4 == 0. */
NSG_API int nsg_gated_activation_forward(const float *x, const float *cond, float *y, int64_t M, int32_t C,
                                         int64_t rows_per_clip, void *stream);
/* dx [M][2C] given dy [M][C] (the gradient w.r.t. cond is the per-clip column sum of dx: nsg_clip_colsum). */
NSG_API int nsg_gated_activation_backward(const float *x, const float *cond, const float *dy, float *dx, int64_t M,
                                          int32_t C, int64_t rows_per_clip, void *stream);

/* F.cross_entropy(logits, target) with mean reduction on rows [M][K] (target int64 [M]):
 * loss_out[0] = mean_m( logsumexp(logits[m]) - logits[m][target[m]] ); dlogits (or NULL) = grad_scale * d loss / d logits. */
NSG_API size_t nsg_cross_entropy_workspace_bytes(int64_t M);
NSG_API int nsg_cross_entropy(const float *logits, const int64_t *target, int64_t M, int32_t K, float grad_scale,
                              float *loss_out, float *dlogits, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Mel -> waveform inversion (the epoch loop's audio export)   src/main.py:164-197, src/audio_tacotron.py:99-116,142-153
 * librosa's stft / istft / filters.mel and scipy's lfilter restated; fp32; frame-major spectrograms [B][T][F], F = n_fft/2+1.
 * ------------------------------------------------------------------------------------------- */

/* S[b][t][f] = max(1e-10, sum_m inv_basis[f][m] * 10^((clip(mel[b][m][t],0,max_abs)*(-min_db)/max_abs + min_db + ref_db)/20))^power
 * (audio_tacotron.py:242-248 _denormalize, :223 _db_to_amp, :202-206 _mel_to_linear, :115 S ** power).  mel [B][n_mels][T]. */
NSG_API int nsg_audio_mel_to_linear(const float *mel, const float *inv_basis, float *S, int32_t B, int32_t n_mels, int32_t T,
                                    int32_t F, float min_level_db, float ref_level_db, float max_abs_value, float power,
                                    void *stream);

/* Griffin-Lim (audio_tacotron.py:142-153): y [B][hop*(T-1)] from magnitudes S [B][T][F]; u [B][T][F] are the uniform
 * [0,1) numbers behind the random initial phases exp(2 pi i u).  n_fft in {512, 1024, 2048}, hop divides n_fft. */
NSG_API size_t nsg_audio_griffin_lim_workspace_bytes(int32_t B, int32_t T, int32_t n_fft);
NSG_API int nsg_audio_griffin_lim(const float *S, const float *u, float *y, int32_t B, int32_t T, int32_t n_fft, int32_t hop,
                                  int32_t iters, void *workspace, size_t workspace_bytes, void *stream);

/* X [B][1 + L/hop][F] complex (interleaved re, im) = librosa.stft(y[b], n_fft, hop): centred, reflect-padded, periodic Hann. */
NSG_API int nsg_audio_stft(const float *y, float *X, int32_t B, int32_t L, int32_t n_fft, int32_t hop, void *stream);

/* y[n] = x[n] + k * y[n-1] per clip (scipy.signal.lfilter([1], [1, -k], x); audio_tacotron.py:28-31), out of place,
 * |k| < 1.  Chunked with a discarded warm-up of ceil(log 1e-9 / log |k|) samples: chunks are independent to below fp32 rounding. */
NSG_API int nsg_audio_inv_preemphasis(const float *x, float *y, int32_t B, int32_t L, float k, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NSG_H_ */
