/*
 * ORACLE (test infrastructure only -- never linked or called by the product path).
 *
 * Plain-C restatement of the reference's nearest-codebook search
 *   /root/reference/src/vector_quantization.py:6-23   (VectorQuantization.forward)
 * including the floating-point evaluation ORDER of the CPU PyTorch ops it calls, so that the
 * int64 indices are reproduced bit-for-bit (SURVEY.md section 8a note 3):
 *
 *   :12  codebook_sqr = torch.sum(codebook ** 2, dim=1)          -> nsg_oracle_rowsumsq()
 *   :13  inputs_sqr   = torch.sum(inputs_flatten ** 2, dim=1)    -> nsg_oracle_rowsumsq()
 *   :16  distances = addmm(codebook_sqr + inputs_sqr, X, E^T, alpha=-2, beta=1)
 *                                                                -> fl(fl(c2+x2) - 2*dot), dot = fmaf chain
 *   :19  torch.min(distances, dim=1)                             -> first minimal index
 *
 * torch.sum over a contiguous row (ATen SumKernel, 8-lane vectors, 4 interleaved accumulators):
 *   nvec = D/8; groups of 4 vectors feed acc[0..3]; left-over vectors go to acc[0];
 *   acc0 = ((acc0+acc1)+acc2)+acc3 lane-wise; result = scalar tail (elements >= 8*nvec) summed
 *   first, then the 8 lanes of acc0 added in lane order.
 * The squares are rounded to fp32 before they are summed (pow is its own op) -- no FMA.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fmaf() must be a real fused op).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#if defined(__GNUC__)
#define NSG_ORACLE_API __attribute__((visibility("default")))
#else
#define NSG_ORACLE_API
#endif

NSG_ORACLE_API void nsg_oracle_rowsumsq(const float *v, int64_t rows, int64_t D, float *out)
{
    for (int64_t r = 0; r < rows; ++r) {
        const float *p = v + r * D;
        volatile float acc[4][8];
        for (int j = 0; j < 4; ++j)
            for (int l = 0; l < 8; ++l) acc[j][l] = 0.0f;
        const int64_t nvec = D / 8;
        const int64_t ngrp = nvec / 4;
        for (int64_t c = 0; c < ngrp; ++c)
            for (int j = 0; j < 4; ++j)
                for (int l = 0; l < 8; ++l) {
                    volatile float sq = p[(4 * c + j) * 8 + l] * p[(4 * c + j) * 8 + l];
                    acc[j][l] = acc[j][l] + sq;
                }
        for (int64_t i = ngrp * 4; i < nvec; ++i)
            for (int l = 0; l < 8; ++l) {
                volatile float sq = p[i * 8 + l] * p[i * 8 + l];
                acc[0][l] = acc[0][l] + sq;
            }
        volatile float fin = 0.0f;
        for (int64_t k = nvec * 8; k < D; ++k) {
            volatile float sq = p[k] * p[k];
            fin = fin + sq;
        }
        for (int l = 0; l < 8; ++l) {
            volatile float a = acc[0][l] + acc[1][l];
            a = a + acc[2][l];
            a = a + acc[3][l];
            fin = fin + a;
        }
        out[r] = fin;
    }
}

/* dist_min may be NULL.  idx is int64 like torch.min's indices. */
NSG_ORACLE_API void nsg_oracle_vq_forward(const float *x, const float *e, int64_t N, int64_t D, int64_t K,
                                          const float *x2, const float *c2, int64_t *idx, float *dist_min)
{
    for (int64_t i = 0; i < N; ++i) {
        const float *xi = x + i * D;
        float best = INFINITY;
        int64_t bk = 0;
        for (int64_t k = 0; k < K; ++k) {
            const float *ek = e + k * D;
            float acc = 0.0f;
            for (int64_t d = 0; d < D; ++d) acc = fmaf(xi[d], ek[d], acc);
            volatile float base = c2[k] + x2[i];
            volatile float dist = base - 2.0f * acc; /* 2*acc is exact: one rounding */
            if (k == 0 || dist < best) { best = dist; bk = k; }
        }
        idx[i] = bk;
        if (dist_min) dist_min[i] = best;
    }
}

/* Full distance matrix (small cases only) for diagnosing a mismatch. */
NSG_ORACLE_API void nsg_oracle_vq_distances(const float *x, const float *e, int64_t N, int64_t D, int64_t K,
                                            const float *x2, const float *c2, float *dist)
{
    for (int64_t i = 0; i < N; ++i)
        for (int64_t k = 0; k < K; ++k) {
            float acc = 0.0f;
            for (int64_t d = 0; d < D; ++d) acc = fmaf(x[i * D + d], e[k * D + d], acc);
            volatile float base = c2[k] + x2[i];
            dist[i * K + k] = base - 2.0f * acc;
        }
}
