/*
 * tome_oracle.c -- CPU restatement of the ToMe merge hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product links, loads or
 * calls this file: only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py may use it, and only as the checker.  The product path
 * (video-how-do-your-tokens-merge_amd/) is HIP-only and fails loudly without
 * its extension.
 *
 * What is restated (reference = sjpollard/video-how-do-your-tokens-merge,
 * paths relative to the reference root):
 *   oracle_match        <- tome/merge.py:17-73   bipartite_soft_matching, index part
 *                          (also :215-251 drop, :274-311 hybrid: same matching)
 *   oracle_merge        <- tome/merge.py:75-85   merge(x, mode) closure
 *   oracle_merge_wavg   <- tome/merge.py:355-369 merge_wavg (x*size, two sum merges, x/size)
 *   oracle_unmerge      <- tome/merge.py:87-100  unmerge(x) closure
 *   oracle_drop         <- tome/merge.py:253-262 drop(x) closure
 *   hybrid pre-scale    <- tome/merge.py:317-326 (dst *= prod of edge flags)
 *
 * Parity pinning: this restatement is checked against golden vectors that
 * tests/golden/generate.py produced by importing the real tome/merge.py in
 * the build container (tests/test_oracle_golden.py).
 *
 * Arithmetic contract (shared bit-for-bit with the HIP kernels):
 *   - every input is first converted to fp32 (exact for bf16/fp16);
 *   - squared norm of a token: channels are taken in blocks of 8 (one 16-byte bf16 lane load);
 *     part_b = fma chain over the block's channels in ascending order starting from 0;
 *     ss = ((part_0 + part_1) + part_2) + ... in ascending block order (fp32 adds);
 *     norm = sqrtf(ss) (correctly rounded); unit[k] = v[k] / norm (IEEE division,
 *     no epsilon: a zero token gives NaN like merge.py:51);
 *   - similarity: acc = 0; for k ascending: acc = fmaf(a[k], b[k], acc)  -- this is
 *     exactly what v_mfma_f32_32x32x2_f32 computes on gfx950 (k-ordered fma chain);
 *   - row max: first maximal column wins; a NaN score wins over any number and the
 *     first NaN keeps the row (torch.max on CPU);
 *   - ranking of rows by node_max, descending; equal keys keep ascending row order
 *     (the reference's argsort is unstable, so ties are undefined there -- SURVEY 7.1);
 *     NaN ranks first, -0 == +0;
 *   - weighted average: p = x*s rounded to fp32, then sequential fp32 adds starting from
 *     the destination's own term and continuing in src_idx (rank) order, then one IEEE
 *     division by the equally accumulated size (merge.py:365-368 executed in fp32).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

enum { MODE_SUM = 0, MODE_MEAN = 1, MODE_AMAX = 2, MODE_PROD = 3, MODE_AMIN = 4 };

/* merge.py:36-44 -- clamp r to half of the unprotected tokens */
ORACLE_API int64_t oracle_effective_r(int64_t T, int64_t r, int class_token, int distill_token) {
    int64_t protected_ = (class_token ? 1 : 0) + (distill_token ? 1 : 0);
    int64_t cap = (T - protected_) / 2;
    if (T - protected_ < 0) cap = -((protected_ - T + 1) / 2); /* python floor division */
    int64_t re = r < cap ? r : cap;
    return re < 0 ? 0 : re;
}

/* merge.py:51 -- metric / metric.norm(dim=-1, keepdim=True), fixed summation order */
static void unit_rows(const float *m, int64_t rows, int64_t D, float *out) {
    for (int64_t t = 0; t < rows; ++t) {
        const float *v = m + t * D;
        float ss = 0.0f;
        for (int64_t k0 = 0; k0 < D; k0 += 8) { /* blocks of 8 channels, see the contract above */
            float part = 0.0f;
            for (int64_t k = k0; k < D && k < k0 + 8; ++k) part = fmaf(v[k], v[k], part);
            ss = ss + part;
        }
        float nrm = sqrtf(ss);
        for (int64_t k = 0; k < D; ++k) out[t * D + k] = v[k] / nrm;
    }
}

static inline int key_before(float ka, int64_t ia, float kb, int64_t ib) {
    /* 1 if (ka, ia) comes before (kb, ib): descending key, NaN first, ties by row. */
    int na = isnan(ka), nb = isnan(kb);
    if (na || nb) {
        if (na && nb) return ia < ib;
        return na;
    }
    if (ka > kb) return 1;
    if (ka < kb) return 0;
    return ia < ib; /* equal, also -0 == +0 */
}

/*
 * oracle_match: metric [n,T,D] fp32 -> index tensors of merge.py:64-73.
 *   src_idx, dst_idx: [n, r_eff]; unm_idx: [n, T1 - r_eff]; node_max [n,T1], node_idx [n,T1]
 *   (either may be NULL).  Returns r_eff (0 => the do_nothing case of merge.py:46-47 and
 *   nothing is written).
 */
ORACLE_API int64_t oracle_match(const float *metric, int64_t n, int64_t T, int64_t D, int64_t r,
                                int class_token, int distill_token, int64_t *src_idx,
                                int64_t *dst_idx, int64_t *unm_idx, float *node_max_out,
                                int32_t *node_idx_out) {
    int64_t re = oracle_effective_r(T, r, class_token, distill_token);
    if (re <= 0) return 0;
    int64_t T1 = (T + 1) / 2, T2 = T / 2;
    float *unit = (float *)malloc(sizeof(float) * (size_t)(T * D));
    float *nmax = (float *)malloc(sizeof(float) * (size_t)T1);
    int32_t *nidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)T1);
    int64_t *rank_of = (int64_t *)malloc(sizeof(int64_t) * (size_t)T1);
    for (int64_t g = 0; g < n; ++g) {
        unit_rows(metric + g * T * D, T, D, unit);
        /* merge.py:52-53,59-64: a = even tokens, b = odd tokens, scores = a @ b^T, row max */
        for (int64_t i = 0; i < T1; ++i) {
            const float *a = unit + (2 * i) * D;
            float best = 0.0f;
            int32_t bj = 0;
            for (int64_t j = 0; j < T2; ++j) {
                const float *b = unit + (2 * j + 1) * D;
                float acc = 0.0f;
                for (int64_t k = 0; k < D; ++k) acc = fmaf(a[k], b[k], acc);
                if (class_token && i == 0) acc = -INFINITY;   /* :59-60 */
                if (distill_token && j == 0) acc = -INFINITY; /* :61-62 */
                if (j == 0 || (!isnan(best) && (isnan(acc) || acc > best))) {
                    best = acc;
                    bj = (int32_t)j;
                }
            }
            nmax[i] = best;
            nidx[i] = bj;
        }
        /* merge.py:65 argsort(descending) with the stable tie rule of the contract */
        for (int64_t i = 0; i < T1; ++i) {
            int64_t c = 0;
            for (int64_t j = 0; j < T1; ++j)
                if (j != i && key_before(nmax[j], j, nmax[i], i)) ++c;
            rank_of[i] = c;
        }
        int64_t *src = src_idx + g * re, *dst = dst_idx + g * re, *unm = unm_idx + g * (T1 - re);
        for (int64_t i = 0; i < T1; ++i) {
            int64_t rk = rank_of[i];
            if (rk < re) { /* :68-69 */
                src[rk] = i;
                dst[rk] = nidx[i];
            } else if (!class_token) { /* :67 */
                unm[rk - re] = i;
            }
        }
        if (class_token) { /* :71-73 unm_idx.sort(): ascending row order */
            int64_t w = 0;
            for (int64_t i = 0; i < T1; ++i)
                if (rank_of[i] >= re) unm[w++] = i;
        }
        if (node_max_out) memcpy(node_max_out + g * T1, nmax, sizeof(float) * (size_t)T1);
        if (node_idx_out) memcpy(node_idx_out + g * T1, nidx, sizeof(int32_t) * (size_t)T1);
    }
    free(unit);
    free(nmax);
    free(nidx);
    free(rank_of);
    return re;
}

/* Same selection, but from a caller-supplied score matrix [n,T1,T2] (merge.py:54-57,
 * the random_merge / random_drop modes whose scores come from torch.rand). */
ORACLE_API int64_t oracle_match_scores(const float *scores, int64_t n, int64_t T, int64_t r,
                                       int class_token, int distill_token, int64_t *src_idx,
                                       int64_t *dst_idx, int64_t *unm_idx, float *node_max_out) {
    int64_t re = oracle_effective_r(T, r, class_token, distill_token);
    if (re <= 0) return 0;
    int64_t T1 = (T + 1) / 2, T2 = T / 2;
    float *nmax = (float *)malloc(sizeof(float) * (size_t)T1);
    int32_t *nidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)T1);
    int64_t *rank_of = (int64_t *)malloc(sizeof(int64_t) * (size_t)T1);
    for (int64_t g = 0; g < n; ++g) {
        for (int64_t i = 0; i < T1; ++i) {
            float best = 0.0f;
            int32_t bj = 0;
            for (int64_t j = 0; j < T2; ++j) {
                float acc = scores[(g * T1 + i) * T2 + j];
                if (class_token && i == 0) acc = -INFINITY;
                if (distill_token && j == 0) acc = -INFINITY;
                if (j == 0 || (!isnan(best) && (isnan(acc) || acc > best))) {
                    best = acc;
                    bj = (int32_t)j;
                }
            }
            nmax[i] = best;
            nidx[i] = bj;
        }
        for (int64_t i = 0; i < T1; ++i) {
            int64_t c = 0;
            for (int64_t j = 0; j < T1; ++j)
                if (j != i && key_before(nmax[j], j, nmax[i], i)) ++c;
            rank_of[i] = c;
        }
        int64_t *src = src_idx + g * re, *dst = dst_idx + g * re, *unm = unm_idx + g * (T1 - re);
        for (int64_t i = 0; i < T1; ++i) {
            int64_t rk = rank_of[i];
            if (rk < re) {
                src[rk] = i;
                dst[rk] = nidx[i];
            } else if (!class_token) {
                unm[rk - re] = i;
            }
        }
        if (class_token) {
            int64_t w = 0;
            for (int64_t i = 0; i < T1; ++i)
                if (rank_of[i] >= re) unm[w++] = i;
        }
        if (node_max_out) memcpy(node_max_out + g * T1, nmax, sizeof(float) * (size_t)T1);
    }
    free(nmax);
    free(nidx);
    free(rank_of);
    return re;
}

/* metric = k.mean(1) of the patches (videomae.py:72-73 ...): keys [n,H,T,D] fp32 -> out [n,T,D];
 * fp32 sum in head order, one division by H (what torch's CPU mean does).  The caller rounds the result to
 * the keys' dtype when that is a 16-bit format. */
ORACLE_API void oracle_head_mean(const float *keys, int64_t n, int64_t H, int64_t T, int64_t D, float *out) {
    for (int64_t g = 0; g < n; ++g)
        for (int64_t t = 0; t < T; ++t)
            for (int64_t k = 0; k < D; ++k) {
                float acc = 0.0f;
                for (int64_t h = 0; h < H; ++h) acc = acc + keys[((g * H + h) * T + t) * D + k];
                out[(g * T + t) * D + k] = acc / (float)H;
            }
}

/* Output row of the merged sequence -> position, honouring the distill layout of
 * merge.py:82-83: [unm[0], dst[0], unm[1:], dst[1:]]. */
static inline int64_t out_row_unm(int64_t k, int distill) { return (distill && k >= 1) ? k + 1 : k; }
static inline int64_t out_row_dst(int64_t j, int64_t U, int distill) {
    if (!distill) return U + j;
    return j == 0 ? 1 : U + j;
}

static inline float reduce_step(float acc, float v, int mode) {
    switch (mode) {
    case MODE_SUM:
    case MODE_MEAN: return acc + v;
    case MODE_PROD: return acc * v;
    case MODE_AMAX: return (isnan(acc) || !(v <= acc)) ? (isnan(acc) ? acc : v) : acc;
    case MODE_AMIN: return (isnan(acc) || !(v >= acc)) ? (isnan(acc) ? acc : v) : acc;
    }
    return acc;
}

/*
 * oracle_merge: x [n,T,C] fp32 -> out [n,T-r,C]   (merge.py:75-85).
 *   edge_keep: NULL, or [n,r] 0/1 flags of the hybrid variant (merge.py:326): a destination
 *   row is first multiplied by the product of the flags of its incoming edges.
 */
ORACLE_API void oracle_merge(const float *x, int64_t n, int64_t T, int64_t C, int64_t r,
                             const int64_t *src_idx, const int64_t *dst_idx,
                             const int64_t *unm_idx, int distill, int mode,
                             const uint8_t *edge_keep, float *out) {
    int64_t T1 = (T + 1) / 2, T2 = T / 2, U = T1 - r, To = T - r;
    float *acc = (float *)malloc(sizeof(float) * (size_t)C);
    for (int64_t g = 0; g < n; ++g) {
        const float *xg = x + g * T * C;
        float *og = out + g * To * C;
        const int64_t *src = src_idx + g * r, *dst = dst_idx + g * r, *unm = unm_idx + g * U;
        for (int64_t k = 0; k < U; ++k) /* :78 gather of the unmerged even tokens */
            memcpy(og + out_row_unm(k, distill) * C, xg + (2 * unm[k]) * C, sizeof(float) * (size_t)C);
        for (int64_t j = 0; j < T2; ++j) { /* :80 scatter_reduce, include_self */
            memcpy(acc, xg + (2 * j + 1) * C, sizeof(float) * (size_t)C);
            int64_t cnt = 1;
            if (edge_keep) {
                for (int64_t k = 0; k < r; ++k)
                    if (dst[k] == j) {
                        float f = edge_keep[g * r + k] ? 1.0f : 0.0f;
                        for (int64_t c = 0; c < C; ++c) acc[c] = acc[c] * f;
                    }
            }
            for (int64_t k = 0; k < r; ++k)
                if (dst[k] == j) {
                    const float *s = xg + (2 * src[k]) * C;
                    for (int64_t c = 0; c < C; ++c) acc[c] = reduce_step(acc[c], s[c], mode);
                    ++cnt;
                }
            if (mode == MODE_MEAN && cnt > 1) {
                float fc = (float)cnt;
                for (int64_t c = 0; c < C; ++c) acc[c] = acc[c] / fc;
            }
            memcpy(og + out_row_dst(j, U, distill) * C, acc, sizeof(float) * (size_t)C);
        }
    }
    free(acc);
}

/*
 * oracle_merge_wavg: merge.py:355-369 in fp32.
 *   size may be NULL (=> ones, :362-363).  x_out [n,T-r,C], size_out [n,T-r].
 */
ORACLE_API void oracle_merge_wavg(const float *x, const float *size, int64_t n, int64_t T,
                                  int64_t C, int64_t r, const int64_t *src_idx,
                                  const int64_t *dst_idx, const int64_t *unm_idx, int distill,
                                  const uint8_t *edge_keep, float *x_out, float *size_out) {
    int64_t T1 = (T + 1) / 2, T2 = T / 2, U = T1 - r, To = T - r;
    float *acc = (float *)malloc(sizeof(float) * (size_t)C);
    for (int64_t g = 0; g < n; ++g) {
        const float *xg = x + g * T * C;
        const float *sg = size ? size + g * T : NULL;
        float *og = x_out + g * To * C;
        float *so = size_out + g * To;
        const int64_t *src = src_idx + g * r, *dst = dst_idx + g * r, *unm = unm_idx + g * U;
        for (int64_t k = 0; k < U; ++k) {
            int64_t t = 2 * unm[k];
            float s = sg ? sg[t] : 1.0f;
            float *o = og + out_row_unm(k, distill) * C;
            for (int64_t c = 0; c < C; ++c) o[c] = (xg[t * C + c] * s) / s; /* :365,:368 */
            so[out_row_unm(k, distill)] = s;
        }
        for (int64_t j = 0; j < T2; ++j) {
            int64_t t = 2 * j + 1;
            float s = sg ? sg[t] : 1.0f;
            for (int64_t c = 0; c < C; ++c) acc[c] = xg[t * C + c] * s;
            float ssum = s;
            if (edge_keep) {
                for (int64_t k = 0; k < r; ++k)
                    if (dst[k] == j) {
                        float f = edge_keep[g * r + k] ? 1.0f : 0.0f;
                        for (int64_t c = 0; c < C; ++c) acc[c] = acc[c] * f;
                        ssum = ssum * f;
                    }
            }
            for (int64_t k = 0; k < r; ++k)
                if (dst[k] == j) {
                    int64_t ts = 2 * src[k];
                    float s2 = sg ? sg[ts] : 1.0f;
                    for (int64_t c = 0; c < C; ++c) acc[c] = acc[c] + xg[ts * C + c] * s2;
                    ssum = ssum + s2;
                }
            float *o = og + out_row_dst(j, U, distill) * C;
            for (int64_t c = 0; c < C; ++c) o[c] = acc[c] / ssum;
            so[out_row_dst(j, U, distill)] = ssum;
        }
    }
    free(acc);
}

/* oracle_unmerge: x [n,T-r,C] -> out [n,T,C]   (merge.py:87-100) */
ORACLE_API void oracle_unmerge(const float *x, int64_t n, int64_t T, int64_t C, int64_t r,
                               const int64_t *src_idx, const int64_t *dst_idx,
                               const int64_t *unm_idx, float *out) {
    int64_t T1 = (T + 1) / 2, T2 = T / 2, U = T1 - r, To = T - r;
    for (int64_t g = 0; g < n; ++g) {
        const float *xg = x + g * To * C;
        float *og = out + g * T * C;
        const int64_t *src = src_idx + g * r, *dst = dst_idx + g * r, *unm = unm_idx + g * U;
        memset(og, 0, sizeof(float) * (size_t)(T * C));
        for (int64_t j = 0; j < T2; ++j) /* :96 */
            memcpy(og + (2 * j + 1) * C, xg + (U + j) * C, sizeof(float) * (size_t)C);
        for (int64_t k = 0; k < U; ++k) /* :97 */
            memcpy(og + (2 * unm[k]) * C, xg + k * C, sizeof(float) * (size_t)C);
        for (int64_t k = 0; k < r; ++k) /* :92,:98 */
            memcpy(og + (2 * src[k]) * C, xg + (U + dst[k]) * C, sizeof(float) * (size_t)C);
    }
}

/* oracle_drop: x [n,T,C] -> out [n,T-r,C]   (merge.py:253-262) */
ORACLE_API void oracle_drop(const float *x, int64_t n, int64_t T, int64_t C, int64_t r,
                            const int64_t *und_idx, int distill, float *out) {
    int64_t T1 = (T + 1) / 2, T2 = T / 2, U = T1 - r, To = T - r;
    for (int64_t g = 0; g < n; ++g) {
        const float *xg = x + g * T * C;
        float *og = out + g * To * C;
        const int64_t *und = und_idx + g * U;
        for (int64_t k = 0; k < U; ++k)
            memcpy(og + out_row_unm(k, distill) * C, xg + (2 * und[k]) * C, sizeof(float) * (size_t)C);
        for (int64_t j = 0; j < T2; ++j)
            memcpy(og + out_row_dst(j, U, distill) * C, xg + (2 * j + 1) * C, sizeof(float) * (size_t)C);
    }
}
