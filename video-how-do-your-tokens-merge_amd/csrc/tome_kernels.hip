// tome_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ToMe merge path and the
// C ABI declared in include/tome_hip.h.  Built with: hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off -shared -fPIC (csrc/build.py).  No torch, no CUDA, no portability layer.
//
// One translation unit: tome_common.h (types), tome_match.h and tome_merge.h (kernels), this file (host).
//
// Launch sequence of one matching (tome_match / tome_match_keys), kernels in tome_match.h:
//   k_unit_rows[_heads]  keys -> fp32 unit vectors, even/odd split, MFMA-fragment order (HBM bound)
//   k_scores_rowmax      A.B^T tile by tile on v_mfma_f32_32x32x2_f32, running row max/argmax in
//                        registers; the [T1,T2] score matrix never exists in memory      (MFMA bound)
//   k_rank_select        folds the j-parts, stable descending rank by counting, writes src/dst/unm
//   k_compact_unm        class-token case only: unm_idx in ascending row order (merge.py:71-73)
// and of one merge (tome_merge_wavg[_regrouped] / tome_merge / tome_drop / tome_unmerge), tome_merge.h:
//   k_merge_rows_fast    streaming waves (2-4 output rows each) + edge waves (rows that receive sources);
//                        <LN>: residual add in front and LayerNorm behind fused in (tome_merge_wavg_ln)
//   k_add_ln_rows        second residual + the next block's first LayerNorm (tome_add_layernorm)
//   k_merge_rows / k_unmerge_rows   generic one-wave-per-row forms                        (HBM bound)
//
// The arithmetic contract (summation orders, tie rules) is the one written at the top of
// oracle/tome_oracle.c; the kernels reproduce it bit for bit.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/tome_hip.h"
#include "tome_common.h"
#include "tome_match.h"
#include "tome_match_filter.h"
#include "tome_merge.h"
#include "tome_attn.h"
#include "tome_attn_stream.h"
#include "tome_attn_resident.h"
#include "tome_embed.h"

// ------------------------------------------------------------------------------------------------
// host side: argument checks, workspace carving, launches
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TOME_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return TOME_OK;
}

extern "C" int tome_abi_version(void) { return TOME_ABI_VERSION; }
extern "C" const char *tome_last_error(void) { return g_err; }

extern "C" int64_t tome_effective_r(int64_t T, int64_t r, int class_token, int distill_token) {
    int64_t prot = (class_token ? 1 : 0) + (distill_token ? 1 : 0);
    int64_t avail = T - prot;
    int64_t cap = avail >= 0 ? avail / 2 : -((-avail + 1) / 2);  // python floor division
    int64_t re = r < cap ? r : cap;
    return re < 0 ? 0 : re;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Per-stage timing of tome_match for bench.py's roofline figures: MEASUREMENT BUILD ONLY (-DTOME_PROFILE_HOOKS ->
// lib/libtome_hip_prof.so, loaded by bench.py's stage-timing leg alone; csrc/build.py).  The product library carries
// neither the entry points nor the repeated launches.
#ifdef TOME_PROFILE_HOOKS
// Events are created when profiling is switched on, never inside a launch path.
#define PROF_EVENTS 4
static thread_local struct {
    bool on = false;
    bool valid = false;
    int reps = 1;  // each stage kernel is launched this many times between its two events (idempotent kernels)
    hipEvent_t ev[PROF_EVENTS];
} g_prof;

static inline void prof_mark(int i, hipStream_t st) {
    if (g_prof.on) (void)hipEventRecord(g_prof.ev[i], st);
}

extern "C" int tome_profile_enable(int on) {
    if (on && !g_prof.on) {
        for (int i = 0; i < PROF_EVENTS; ++i)
            if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return fail(TOME_ELAUNCH, "hipEventCreate failed");
        g_prof.on = true;
        g_prof.valid = false;
    }
    if (on) g_prof.reps = on;
    if (!on && g_prof.on) {
        for (int i = 0; i < PROF_EVENTS; ++i) (void)hipEventDestroy(g_prof.ev[i]);
        g_prof.on = false;
        g_prof.valid = false;
        g_prof.reps = 1;
    }
    return TOME_OK;
}

extern "C" int tome_profile_read(float *stage_ms, int max_stages) {
    if (!g_prof.on || !g_prof.valid || !stage_ms) return fail(TOME_EINVAL, "tome_profile_read: no profiled call");
    if (hipEventSynchronize(g_prof.ev[PROF_EVENTS - 1]) != hipSuccess)
        return fail(TOME_ELAUNCH, "tome_profile_read: event synchronize failed");
    for (int i = 0; i + 1 < PROF_EVENTS && i < max_stages; ++i) {
        if (hipEventElapsedTime(&stage_ms[i], g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess)
            return fail(TOME_ELAUNCH, "tome_profile_read: elapsed time failed");
        stage_ms[i] /= (float)g_prof.reps;
    }
    return TOME_OK;
}

static inline int prof_reps_now() { return g_prof.on ? g_prof.reps : 1; }
static inline void prof_done(int rc) { g_prof.valid = g_prof.on && rc == TOME_OK; }
#else
static inline void prof_mark(int, hipStream_t) {}
static inline int prof_reps_now() { return 1; }
static inline void prof_done(int) {}
#endif

#ifdef TOME_DIAG_CLOCK
// diagnostic build: in-kernel clock of the last k_scores_rowmax launch = sum(cycles) / sum(100 MHz ticks) over
// its waves, in GHz; also the mean wave duration in microseconds
extern "C" int tome_diag_clock(double *ghz, double *wave_us, int64_t *waves) {
    static unsigned long long host[TOME_DIAG_SLOTS * 2];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_diag_stamps), sizeof(host)) != hipSuccess)
        return fail(TOME_ELAUNCH, "tome_diag_clock: copy failed");
    double cyc = 0, ticks = 0;
    int64_t nw = 0;
    for (int i = 0; i < TOME_DIAG_SLOTS; ++i)
        if (host[2 * i + 1] > 0) {
            cyc += (double)host[2 * i];
            ticks += (double)host[2 * i + 1];
            ++nw;
        }
    if (!nw) return fail(TOME_EINVAL, "tome_diag_clock: no stamps");
    *ghz = cyc / ticks * 0.1;
    *wave_us = ticks / nw * 0.01;
    *waves = nw;
    return TOME_OK;
}
#endif

struct MatchWs {
    float *unitA, *unitB, *part_max;
    int *part_idx, *rank;
    uint8_t *badA, *badB;
    int ntA, ntB, nchunk;
    int64_t groupA_f4, groupB_f4;  // float4 per group of each unit set
    // the filter path (tome_match_filter.h; D <= 64): bf16 means in MFMA fragment order, norms, candidate lists
    uint4 *vA, *vB;
    float *normA, *normB, *invB, *node_max;
    int *node_idx;
    CandEntry *cand;
    uint8_t *cand_n, *tile_flag;
    int T2p;
    size_t bytes;
};

static MatchWs carve(void *base, int64_t n, int64_t T, int64_t D) {
    const int64_t T1 = (T + 1) / 2, T2 = T / 2;
    MatchWs w;
    w.nchunk = (int)((D + 63) / 64);
    w.ntA = (int)((T1 + TILE_ROWS - 1) / TILE_ROWS);
    w.ntB = (int)((T2 + TILE_ROWS - 1) / TILE_ROWS);
    w.groupA_f4 = (int64_t)w.ntA * w.nchunk * 512;
    w.groupB_f4 = (int64_t)w.ntB * w.nchunk * 512;
    size_t off = 0;
    char *b = (char *)base;
    w.unitA = (float *)(b + off); off = align_up(off + 16 * (size_t)(n * w.groupA_f4), 256);
    w.unitB = (float *)(b + off); off = align_up(off + 16 * (size_t)(n * w.groupB_f4), 256);
    w.part_max = (float *)(b + off); off = align_up(off + sizeof(float) * (size_t)(n * MAX_WJ * T1), 256);
    w.part_idx = (int *)(b + off); off = align_up(off + sizeof(int) * (size_t)(n * MAX_WJ * T1), 256);
    w.rank = (int *)(b + off); off = align_up(off + sizeof(int) * (size_t)(n * T1), 256);
    w.badA = (uint8_t *)(b + off); off = align_up(off + (size_t)(n * T1), 256);
    w.badB = (uint8_t *)(b + off); off = align_up(off + (size_t)(n * (T2 > 0 ? T2 : 1)), 256);
    w.T2p = w.ntB * TILE_ROWS;
    w.vA = w.vB = nullptr;
    if (w.nchunk == 1) {
        w.vA = (uint4 *)(b + off); off = align_up(off + 4096 * (size_t)(n * w.ntA), 256);
        w.vB = (uint4 *)(b + off); off = align_up(off + 4096 * (size_t)(n * (w.ntB > 0 ? w.ntB : 1)), 256);
        w.normA = (float *)(b + off); off = align_up(off + sizeof(float) * (size_t)(n * T1), 256);
        w.normB = (float *)(b + off); off = align_up(off + sizeof(float) * (size_t)(n * (T2 > 0 ? T2 : 1)), 256);
        w.invB = (float *)(b + off); off = align_up(off + sizeof(float) * (size_t)(n * (w.T2p > 0 ? w.T2p : 1)) + 64, 256);
        w.node_max = (float *)(b + off); off = align_up(off + sizeof(float) * (size_t)(n * T1), 256);
        w.node_idx = (int *)(b + off); off = align_up(off + sizeof(int) * (size_t)(n * T1), 256);
        w.cand = (CandEntry *)(b + off); off = align_up(off + sizeof(CandEntry) * 2 * FILT_KH * (size_t)(n * T1), 256);
        w.cand_n = (uint8_t *)(b + off); off = align_up(off + 2 * (size_t)(n * T1), 256);
        w.tile_flag = (uint8_t *)(b + off); off = align_up(off + (size_t)(n * w.ntA), 256);
    }
    w.bytes = off;
    return w;
}

extern "C" size_t tome_match_workspace_bytes(int64_t n, int64_t T, int64_t D) {
    if (n <= 0 || T <= 0 || D <= 0) return 0;
    return carve(nullptr, n, T, D).bytes;
}

static int launch_select(const MatchWs &w, int nparts, bool nan_flags, int64_t n, int64_t T, int64_t re, int class_token,
                         int distill_token, int64_t *src_idx, int64_t *dst_idx, int64_t *unm_idx,
                         float *node_max, int32_t *row_map, hipStream_t st) {
    const int T1 = (int)((T + 1) / 2);
    dim3 grid((T1 + 63) / 64, (unsigned)n);
    const int quarter = (((T1 + 3) >> 2) + 1) & ~1;
    const size_t lds = sizeof(unsigned long long) * (size_t)(4 * quarter);
    if (lds > 160 * 1024) return fail(TOME_EINVAL, "sequences of more than ~40000 tokens do not fit the ranking kernel's LDS");
    hipLaunchKernelGGL(k_rank_select, grid, dim3(256), lds, st, w.part_max, w.part_idx, nparts, (int)n, T1,
                       (int)(T / 2), nan_flags ? w.badA : nullptr, nan_flags ? w.badB : nullptr, (int)re, class_token,
                       distill_token, src_idx, dst_idx, unm_idx, node_max, w.rank, row_map);
    if (int rc = check_launch("k_rank_select")) return rc;
    if (class_token) {
        hipLaunchKernelGGL(k_compact_unm, dim3((unsigned)n), dim3(256), 0, st, w.rank, T1, (int)re,
                           distill_token, unm_idx, row_map);
        if (int rc = check_launch("k_compact_unm")) return rc;
    }
    return TOME_OK;
}

// The filter path (tome_match_filter.h) serves bf16 metrics of at most 64 channels when the launch has enough A tiles
// to fill the chip with one wave per tile; everything else keeps the fp32 pass.  TOME_SCORES_FILTER=0 switches it off
// (measurement switch, read per call).
static bool use_filter(const MatchWs &w, int dtype, int64_t n, int64_t D) {
    const char *e = getenv("TOME_SCORES_FILTER");
    if (e && e[0] == '0') return false;
    const int64_t min_tiles = (e && e[0] == '2') ? 1 : 1024;  // ("2": also for small launches -- the tests)
    return dtype == TOME_BF16 && w.nchunk == 1 && D % 8 == 0 && w.ntB > 0 && w.T2p <= FILT_MAX_T2P &&
           n * w.ntA >= min_tiles;
}

// shared tail of tome_match / tome_match_keys: stages 2 (similarity + row max) and 3 (rank + select)
static int match_tail(const MatchWs &w, int64_t n, int64_t T, int64_t D, int64_t re, int class_token, int distill_token,
                      int64_t *src_idx, int64_t *dst_idx, int64_t *unm_idx, float *node_max, int32_t *row_map,
                      hipStream_t st, bool filtered) {
    const int T1 = (int)((T + 1) / 2), T2 = (int)(T / 2);
    const int prof_reps = prof_reps_now();
    // 2. similarity + row max/argmax: one single-wave workgroup per (group, A tile, j-part); the B tiles are
    // split into WJ parts so that the launch has >= ~6 waves per SIMD (1024 SIMDs) whatever the batch
    static const long target_waves = [] {
        const char *e = getenv("TOME_SCORES_WAVES");  // tuning knob, default from measurements on MI355X
        long v = e ? atol(e) : 0;
        return v > 0 ? v : 6144L;
    }();
    int WJ = (int)((target_waves + n * w.ntA - 1) / (n * w.ntA));
    if (WJ > MAX_WJ) WJ = MAX_WJ;
    if (WJ > w.ntB) WJ = w.ntB;
    if (WJ < 1) WJ = 1;
    const unsigned nb2 = (unsigned)(((n + 7) / 8) * 8 * w.ntA * WJ);
    for (int rep = 0; rep < prof_reps; ++rep) {
        if (filtered) {
            // 2a. approximate scores on the bf16 matrix pipe, candidate columns per row (tome_match_filter.h)
            hipLaunchKernelGGL(k_scores_filter, dim3((unsigned)(((n + 7) / 8) * 8 * ((w.ntA + FILT_ATW - 1) / FILT_ATW))),
                               dim3(64), sizeof(float) * (size_t)w.T2p, st, w.vA, w.vB,
                               w.normA, w.invB, (int)n, T1, T2, w.T2p, w.ntA, w.ntB, distill_token, w.cand, w.cand_n,
                               w.tile_flag);
            if (int rc = check_launch("k_scores_filter")) return rc;
            // 2b. the exact score of every row's winner; the same launch carries one wave per A tile that takes the
            // fp32 pass when the filter flagged the tile (overflowed candidate list, norm out of range) and leaves at
            // once otherwise
            const int64_t exact_blocks = (n * T1 + 255) / 256, fb_blocks = (n * w.ntA + 3) / 4;
            hipLaunchKernelGGL(k_exact_rows, dim3((unsigned)(exact_blocks + fb_blocks)), dim3(256), 0, st, w.vA, w.vB,
                               w.normA, w.normB, (int)n, T1, T2, (int)D, w.ntA, w.ntB, w.cand, w.cand_n, w.tile_flag,
                               (int)exact_blocks, distill_token, w.node_max, w.node_idx);
            if (int rc = check_launch("k_exact_rows")) return rc;
            continue;
        }
        if (w.nchunk == 1)
            hipLaunchKernelGGL(k_scores_rowmax<true>, dim3(nb2), dim3(64), 0, st, (const f32x4 *)w.unitA,
                               (const f32x4 *)w.unitB, (int)n, T1, T2, w.nchunk, w.ntA, w.ntB, WJ, w.groupA_f4,
                               w.groupB_f4, distill_token, w.part_max, w.part_idx, nullptr);
        else
            hipLaunchKernelGGL(k_scores_rowmax<false>, dim3(nb2), dim3(64), 0, st, (const f32x4 *)w.unitA,
                               (const f32x4 *)w.unitB, (int)n, T1, T2, w.nchunk, w.ntA, w.ntB, WJ, w.groupA_f4,
                               w.groupB_f4, distill_token, w.part_max, w.part_idx, nullptr);
        if (int rc = check_launch("k_scores_rowmax")) return rc;
    }
    prof_mark(2, st);

    // 3. rank + select
    int rc = TOME_OK;
    MatchWs ws = w;
    if (filtered) {
        ws.part_max = w.node_max;
        ws.part_idx = w.node_idx;
    }
    for (int rep = 0; rep < prof_reps && rc == TOME_OK; ++rep)
        rc = launch_select(ws, filtered ? 1 : WJ, true, n, T, re, class_token, distill_token, src_idx, dst_idx, unm_idx, node_max, row_map, st);
    prof_mark(3, st);
    prof_done(rc);
    return rc;
}
extern "C" int tome_match(const void *metric, int dtype, int64_t n, int64_t T, int64_t D, int64_t stride_n,
                          int64_t stride_t, int64_t r, int class_token, int distill_token, int64_t *src_idx,
                          int64_t *dst_idx, int64_t *unm_idx, float *node_max, int32_t *row_map,
                          void *workspace, size_t workspace_bytes, tome_stream_t stream) {
    if (!metric || n <= 0 || T <= 0 || D <= 0) return fail(TOME_EINVAL, "tome_match: bad shape/pointer");
    if (n > 0x7fffffff / T || (int64_t)n * T * ((D + 63) / 64 * 64) > (int64_t)1 << 40)
        return fail(TOME_EINVAL, "tome_match: problem too large");
    const int64_t re = tome_effective_r(T, r, class_token, distill_token);
    if (re <= 0) return TOME_OK;
    if (!src_idx || !dst_idx || (!unm_idx && (T + 1) / 2 > re))
        return fail(TOME_EINVAL, "tome_match: null index buffer");
    if (!workspace || workspace_bytes < tome_match_workspace_bytes(n, T, D))
        return fail(TOME_EWORKSPACE, "tome_match: workspace %zu < %zu bytes", workspace_bytes,
                    tome_match_workspace_bytes(n, T, D));
    if (((uintptr_t)workspace & 255) != 0) return fail(TOME_EINVAL, "tome_match: workspace not 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const MatchWs w = carve(workspace, n, T, D);
    // 1. unit vectors
    prof_mark(0, st);
    const size_t es = dtype == TOME_F32 ? 4 : 2;
    const bool fast = (D % 8 == 0) && (((uintptr_t)metric) % 16 == 0) && ((stride_n * es) % 16 == 0) &&
                      ((stride_t * es) % 16 == 0);
    const bool filtered = fast && use_filter(w, dtype, n, D);
    bool launched = false;
    const int prof_reps = prof_reps_now();
    for (int rep = 0; rep < prof_reps; ++rep) {
    launched = false;
    if (filtered) {
        hipLaunchKernelGGL(k_unit_rows_f<false>, dim3((unsigned)((n * T + 31) / 32)), dim3(256), 0, st,
                           (const bf16_t *)metric, stride_n, 1, (int64_t)0, (int64_t)0, stride_t, (int)n, 1, (int)T, (int)D,
                           w.vA, w.vB, w.ntA, w.ntB, w.normA, w.normB, w.invB, w.T2p, w.badA, w.badB);
        continue;
    }
#define UNIT_FAST(TY, NCH)                                                                                    \
    hipLaunchKernelGGL((k_unit_rows<TY, NCH>), dim3((unsigned)((n * T + 31) / 32)), dim3(256), 0, st,          \
                       (const TY *)metric, stride_n, stride_t, (int)n, (int)T, (int)D, w.unitA, w.unitB,       \
                       w.groupA_f4, w.groupB_f4, w.badA, w.badB);                                              \
    launched = true
#define UNIT_NCH(TY)                                       \
    switch (w.nchunk) {                                    \
    case 1: UNIT_FAST(TY, 1); break;                       \
    case 2: UNIT_FAST(TY, 2); break;                       \
    case 3: UNIT_FAST(TY, 3); break;                       \
    case 4: UNIT_FAST(TY, 4); break;                       \
    case 6: UNIT_FAST(TY, 6); break;                       \
    case 8: UNIT_FAST(TY, 8); break;                       \
    case 12: UNIT_FAST(TY, 12); break;                     \
    case 16: UNIT_FAST(TY, 16); break;                     \
    default: break;                                        \
    }
#define UNIT_GENERIC(TY)                                                                                       \
    hipLaunchKernelGGL((k_unit_rows_generic<TY>), dim3((unsigned)((n * T + 255) / 256)), dim3(256), 0, st,      \
                       (const TY *)metric, stride_n, stride_t, (int)n, (int)T, (int)D, w.nchunk * 64, w.unitA,  \
                       w.unitB, w.groupA_f4, w.groupB_f4, w.badA, w.badB)
    switch (dtype) {
    case TOME_F32:
        if (fast) { UNIT_NCH(float) }
        if (!launched) UNIT_GENERIC(float);
        break;
    case TOME_BF16:
        if (fast) { UNIT_NCH(bf16_t) }
        if (!launched) UNIT_GENERIC(bf16_t);
        break;
    case TOME_F16:
        if (fast) { UNIT_NCH(f16_t) }
        if (!launched) UNIT_GENERIC(f16_t);
        break;
    default: return fail(TOME_EINVAL, "tome_match: dtype %d", dtype);
    }
#undef UNIT_FAST
#undef UNIT_NCH
#undef UNIT_GENERIC
    }
    if (int rc = check_launch("k_unit_rows")) return rc;
    prof_mark(1, st);

    return match_tail(w, n, T, D, re, class_token, distill_token, src_idx, dst_idx, unm_idx, node_max, row_map, st, filtered);
}

extern "C" int tome_match_keys(const void *keys, int dtype, int64_t n, int64_t H, int64_t T, int64_t D,
                               int64_t stride_n, int64_t inner, int64_t stride_inner, int64_t stride_h,
                               int64_t stride_t, int64_t r, int class_token,
                               int distill_token, int64_t *src_idx, int64_t *dst_idx, int64_t *unm_idx,
                               float *node_max, int32_t *row_map, void *workspace, size_t workspace_bytes,
                               tome_stream_t stream) {
    if (!keys || n <= 0 || T <= 0 || H <= 0) return fail(TOME_EINVAL, "tome_match_keys: bad shape/pointer");
    if (D != 64) return fail(TOME_EINVAL, "tome_match_keys: head dimension %lld (only 64 is fused)", (long long)D);
    if (n > 0x7fffffff / T) return fail(TOME_EINVAL, "tome_match_keys: problem too large");
    const size_t es = dtype == TOME_F32 ? 4 : 2;
    if (dtype < TOME_F32 || dtype > TOME_F16) return fail(TOME_EINVAL, "tome_match_keys: dtype %d", dtype);
    if (((uintptr_t)keys) % 16 || (stride_n * es) % 16 || (stride_h * es) % 16 || (stride_t * es) % 16 ||
        (stride_inner * es) % 16)
        return fail(TOME_EINVAL, "tome_match_keys: keys must be 16-byte aligned in every stride");
    if (inner < 1 || n % inner) return fail(TOME_EINVAL, "tome_match_keys: %lld groups do not split into %lld per clip",
                                            (long long)n, (long long)inner);
    const int64_t re = tome_effective_r(T, r, class_token, distill_token);
    if (re <= 0) return TOME_OK;
    if (!src_idx || !dst_idx || (!unm_idx && (T + 1) / 2 > re))
        return fail(TOME_EINVAL, "tome_match_keys: null index buffer");
    if (!workspace || workspace_bytes < tome_match_workspace_bytes(n, T, D))
        return fail(TOME_EWORKSPACE, "tome_match_keys: workspace %zu < %zu bytes", workspace_bytes,
                    tome_match_workspace_bytes(n, T, D));
    if (((uintptr_t)workspace & 255) != 0) return fail(TOME_EINVAL, "tome_match_keys: workspace not 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const MatchWs w = carve(workspace, n, T, D);
    prof_mark(0, st);
    const int prof_reps = prof_reps_now();
    const unsigned nb = (unsigned)((n * T + 31) / 32);
    const bool filtered = use_filter(w, dtype, n, D);
    for (int rep = 0; rep < prof_reps; ++rep) {
        if (filtered) {
            hipLaunchKernelGGL(k_unit_rows_f<true>, dim3(nb), dim3(256), 0, st, (const bf16_t *)keys, stride_n, (int)inner,
                               stride_inner, stride_h, stride_t, (int)n, (int)H, (int)T, (int)D, w.vA, w.vB, w.ntA, w.ntB,
                               w.normA, w.normB, w.invB, w.T2p, w.badA, w.badB);
            continue;
        }
        switch (dtype) {
        case TOME_F32:
            hipLaunchKernelGGL(k_unit_rows_heads<float>, dim3(nb), dim3(256), 0, st, (const float *)keys, stride_n,
                               (int)inner, stride_inner, stride_h, stride_t, (int)n, (int)H, (int)T, w.unitA, w.unitB, w.groupA_f4, w.groupB_f4,
                               w.badA, w.badB);
            break;
        case TOME_BF16:
            hipLaunchKernelGGL(k_unit_rows_heads<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t *)keys, stride_n,
                               (int)inner, stride_inner, stride_h, stride_t, (int)n, (int)H, (int)T, w.unitA, w.unitB, w.groupA_f4, w.groupB_f4,
                               w.badA, w.badB);
            break;
        default:
            hipLaunchKernelGGL(k_unit_rows_heads<f16_t>, dim3(nb), dim3(256), 0, st, (const f16_t *)keys, stride_n,
                               (int)inner, stride_inner, stride_h, stride_t, (int)n, (int)H, (int)T, w.unitA, w.unitB, w.groupA_f4, w.groupB_f4,
                               w.badA, w.badB);
            break;
        }
    }
    if (int rc = check_launch("k_unit_rows_heads")) return rc;
    prof_mark(1, st);
    return match_tail(w, n, T, D, re, class_token, distill_token, src_idx, dst_idx, unm_idx, node_max, row_map, st, filtered);
}

extern "C" int tome_match_scores(const float *scores, int64_t n, int64_t T, int64_t r, int class_token,
                                 int distill_token, int64_t *src_idx, int64_t *dst_idx, int64_t *unm_idx,
                                 float *node_max, int32_t *row_map, void *workspace, size_t workspace_bytes,
                                 tome_stream_t stream) {
    if (!scores || n <= 0 || T <= 0) return fail(TOME_EINVAL, "tome_match_scores: bad shape/pointer");
    const int64_t re = tome_effective_r(T, r, class_token, distill_token);
    if (re <= 0) return TOME_OK;
    if (!src_idx || !dst_idx || (!unm_idx && (T + 1) / 2 > re))
        return fail(TOME_EINVAL, "tome_match_scores: null index buffer");
    if (!workspace || workspace_bytes < tome_match_workspace_bytes(n, T, 1))
        return fail(TOME_EWORKSPACE, "tome_match_scores: workspace too small");
    if (((uintptr_t)workspace & 255) != 0) return fail(TOME_EINVAL, "tome_match_scores: workspace alignment");
    hipStream_t st = (hipStream_t)stream;
    const MatchWs w = carve(workspace, n, T, 1);
    const int T1 = (int)((T + 1) / 2), T2 = (int)(T / 2);
    const unsigned nb = (unsigned)((n * T1 + 3) / 4);
    hipLaunchKernelGGL(k_rowmax_given, dim3(nb), dim3(256), 0, st, scores, (int)n, T1, T2, class_token,
                       distill_token, w.part_max, w.part_idx);
    if (int rc = check_launch("k_rowmax_given")) return rc;
    return launch_select(w, 1, false, n, T, re, class_token, distill_token, src_idx, dst_idx, unm_idx, node_max, row_map,
                         st);
}

extern "C" int tome_edge_keep(const float *node_max, const int64_t *src_idx, int64_t n, int64_t T, int64_t r,
                              float threshold, uint8_t *edge_keep, tome_stream_t stream) {
    if (!node_max || !src_idx || !edge_keep || n <= 0 || r <= 0) return fail(TOME_EINVAL, "tome_edge_keep: bad args");
    const int T1 = (int)((T + 1) / 2);
    const unsigned nb = (unsigned)((n * r + 255) / 256);
    hipLaunchKernelGGL(k_edge_keep, dim3(nb), dim3(256), 0, (hipStream_t)stream, node_max, src_idx, (int)n, T1,
                       (int)r, threshold, edge_keep);
    return check_launch("k_edge_keep");
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

static TokLayout contiguous_layout(int64_t T, int64_t C) { return TokLayout{0, T * C, 0, C, 1}; }

template <typename TX, typename TS, int OP>
static int launch_merge_rows(const void *x, const void *size, int64_t n, int64_t T, int64_t C, int64_t r,
                             const int64_t *src, const int64_t *dst, const int64_t *unm, int distill,
                             const uint8_t *keep, void *xout, void *sout, hipStream_t st,
                             const TokLayout *lin_p = nullptr, const TokLayout *lout_p = nullptr, int cls_rows = 0,
                             const LnArgs *ln_p = nullptr, void *lsout = nullptr) {
    constexpr int VEC = 16 / sizeof(TX);
    const int64_t To = T - r;
    const TokLayout lin = lin_p ? *lin_p : contiguous_layout(T, C);
    const TokLayout lout = lout_p ? *lout_p : contiguous_layout(To, C);
    const bool vec_ok = (C % VEC == 0) && aligned16(x) && aligned16(xout);
    const int64_t cpr = C / VEC;  // 16-byte chunks per row
    const LnArgs no_ln{nullptr, nullptr, nullptr, 0.0f, nullptr, 0, TokLayout{0, 0, 0, 0, 1}, nullptr, 0, nullptr};
    if (vec_ok && cpr <= FAST_NIT * WAVE) {
        // rows per wave: measured on MI355X, NIT=6 (four 1536-byte rows per wave for 768-channel bf16 tokens)
        // beats NIT=3 by ~4 %; TOME_MERGE_NIT=3 keeps the other variant reachable for re-measurement
        static const int nit_pref = [] {
            const char *e = getenv("TOME_MERGE_NIT");
            int v = e ? atoi(e) : 0;
            return (v == 3 || v == 6) ? v : 6;
        }();
        // with the LayerNorm fused: 6 chunks per lane for x alone; with the residual stream next to it 3 per lane
        // (77 instead of 107 VGPRs, six instead of four waves per SIMD) measured 5 % faster (124 vs 131 us)
        static const int nit_ln = [] {
            const char *e = getenv("TOME_MERGE_LN_NIT");
            int v = e ? atoi(e) : 0;
            return (v == 3 || v == 6) ? v : 0;
        }();
        const int nit = (cpr <= 3 * WAVE) ? (ln_p ? (nit_ln ? nit_ln : (ln_p->addend ? 3 : 6)) : nit_pref) : FAST_NIT;
        int R = (int)((nit * WAVE) / cpr);
        if (R > FAST_MAXR) R = FAST_MAXR;
        // grid.x = the blocks of one group (streaming waves, then the edge waves), (y, z) = group (+ rows of blocks
        // for the class tokens behind them)
        const int64_t bpg = ((To + R - 1) / R + (OP == OP_DROP ? 0 : r) + 3) / 4;
        const int64_t ny = n + (cls_rows ? (cls_rows + 4 * bpg - 1) / (4 * bpg) : 0);
        const int64_t gy = ny < 65535 ? ny : 65535, gz = (ny + gy - 1) / gy;
        if (gz > 65535) return fail(TOME_EINVAL, "merge: too many groups (%lld)", (long long)n);
        dim3 grid((unsigned)bpg, (unsigned)gy, (unsigned)gz);
        // XCD-aware numbering of the workgroups (MergeSched, csrc/tome_merge.h) for the launches where many destinations
        // receive sources (8 r >= T: TimeSformer / Motionformer frame groups at r = 32, late layers at r = 16) -- there the
        // edge blocks at the end of every group otherwise land on the same XCDs in every group; measured per layer with
        // tools/regroup_kernel_times.py: -4 ... -7 % at r = 32, +-1 % at r = 16, +3 % at r = 8 and +3.6 % on the
        // benchmark's VideoMAE launches (598 vs 577 us), hence not there.  TOME_MERGE_XCD=0 / 1 forces it off / on
        // (measurement switch, read per call).
        MergeSched sch{(unsigned)bpg, (unsigned)(bpg * ny), 0u, 0, 0ull};
        {
            const char *xe = getenv("TOME_MERGE_XCD");
            const int64_t total = bpg * ny;
            const bool want = xe ? xe[0] == '1' : (OP != OP_DROP && r <= 64 && 8 * r >= T);
            if (want && total < (1ll << 28) && bpg < (1ll << 12) && total >= 64) {
                sch.per_xcd = (unsigned)((total + 7) / 8);
                sch.on = 1;
                sch.magic = ((1ull << 40) + (unsigned long long)bpg - 1ull) / (unsigned long long)bpg;
                grid = dim3(8u * sch.per_xcd, 1u, 1u);
            }
        }
#ifdef TOME_OCC_PROBE
        const char *lds_e = getenv("TOME_MERGE_LDS");
        const unsigned occ_lds = lds_e ? (unsigned)atoi(lds_e) : 0u;
#define OCC_LDS occ_lds
#else
#define OCC_LDS 0
#endif
        if (ln_p) {
            if (OP != OP_WAVG || sizeof(TX) != 2 || cpr > 2 * WAVE || !aligned16(ln_p->y) || !aligned16(ln_p->weight) ||
                !aligned16(ln_p->bias))
                return fail(TOME_EINVAL, "fused LayerNorm needs 16-bit tokens with C <= 1024 and 16-byte aligned buffers");
            if constexpr (OP == OP_WAVG && sizeof(TX) == 2) {
                // many destinations receive sources: the streaming waves skip those rows (EAGER, csrc/tome_merge.h);
                // TOME_MERGE_EAGER=0 / 1 forces it off / on where r <= 64 (measurement switch, read per call)
                const char *ee = getenv("TOME_MERGE_EAGER");
                const bool eager = r <= 64 && ((ee && ee[0] == '1') || (!(ee && ee[0] == '0') && 8 * r >= T));
                if (eager && nit == 3)
                    hipLaunchKernelGGL((k_merge_rows_fast<TX, TS, OP, 3, true, true>), grid, dim3(256), OCC_LDS, st, (const TX *)x,
                                       (const TS *)size, (int)n, (int)T, (int)C, (int)r, R, (int)cpr, (int)((To + R - 1) / R), src, dst, unm,
                                       distill, keep, (TX *)xout, (TS *)sout, lin, lout, cls_rows, *ln_p, (TS *)lsout, sch);
                else if (eager)
                    hipLaunchKernelGGL((k_merge_rows_fast<TX, TS, OP, 6, true, true>), grid, dim3(256), OCC_LDS, st, (const TX *)x,
                                       (const TS *)size, (int)n, (int)T, (int)C, (int)r, R, (int)cpr, (int)((To + R - 1) / R), src, dst, unm,
                                       distill, keep, (TX *)xout, (TS *)sout, lin, lout, cls_rows, *ln_p, (TS *)lsout, sch);
                else if (nit == 3)
                    hipLaunchKernelGGL((k_merge_rows_fast<TX, TS, OP, 3, true>), grid, dim3(256), OCC_LDS, st, (const TX *)x,
                                       (const TS *)size, (int)n, (int)T, (int)C, (int)r, R, (int)cpr, (int)((To + R - 1) / R), src, dst, unm,
                                       distill, keep, (TX *)xout, (TS *)sout, lin, lout, cls_rows, *ln_p, (TS *)lsout, sch);
                else
                    hipLaunchKernelGGL((k_merge_rows_fast<TX, TS, OP, 6, true>), grid, dim3(256), OCC_LDS, st, (const TX *)x,
                                       (const TS *)size, (int)n, (int)T, (int)C, (int)r, R, (int)cpr, (int)((To + R - 1) / R), src, dst, unm,
                                       distill, keep, (TX *)xout, (TS *)sout, lin, lout, cls_rows, *ln_p, (TS *)lsout, sch);
            }
        } else if (nit == 3)
            hipLaunchKernelGGL((k_merge_rows_fast<TX, TS, OP, 3>), grid, dim3(256), OCC_LDS, st, (const TX *)x,
                               (const TS *)size, (int)n, (int)T, (int)C, (int)r, R, (int)cpr, (int)((To + R - 1) / R), src, dst, unm, distill,
                               keep, (TX *)xout, (TS *)sout, lin, lout, cls_rows, no_ln, (TS *)lsout, sch);
        else
            hipLaunchKernelGGL((k_merge_rows_fast<TX, TS, OP, 6>), grid, dim3(256), OCC_LDS, st, (const TX *)x,
                               (const TS *)size, (int)n, (int)T, (int)C, (int)r, R, (int)cpr, (int)((To + R - 1) / R), src, dst, unm, distill,
                               keep, (TX *)xout, (TS *)sout, lin, lout, cls_rows, no_ln, (TS *)lsout, sch);
        return check_launch("k_merge_rows_fast");
    }
    if (cls_rows || ln_p)
        return fail(TOME_EINVAL, "regrouped / LayerNorm-fused merge needs rows of whole 16-byte chunks (C=%lld)", (long long)C);
    const int64_t rows = n * To;
    const unsigned nb = (unsigned)((rows + 3) / 4);
    if (vec_ok)
        hipLaunchKernelGGL((k_merge_rows<TX, TS, VEC, OP>), dim3(nb), dim3(256), 0, st, (const TX *)x,
                           (const TS *)size, (int)n, (int)T, (int)C, (int)r, src, dst, unm, distill, keep, (TX *)xout,
                           (TS *)sout, lin, lout, (TS *)lsout);
    else
        hipLaunchKernelGGL((k_merge_rows<TX, TS, 1, OP>), dim3(nb), dim3(256), 0, st, (const TX *)x,
                           (const TS *)size, (int)n, (int)T, (int)C, (int)r, src, dst, unm, distill, keep, (TX *)xout,
                           (TS *)sout, lin, lout, (TS *)lsout);
    return check_launch("k_merge_rows");
}

static int check_merge_args(const char *who, const void *x, int64_t n, int64_t T, int64_t C, int64_t r,
                            const void *out) {
    if (!x || !out || n <= 0 || T <= 0 || C <= 0) return fail(TOME_EINVAL, "%s: bad shape/pointer", who);
    if (r <= 0 || r > T / 2) return fail(TOME_EINVAL, "%s: r=%lld outside (0, T/2]", who, (long long)r);
    if (n * (T - r) > 0x7fffffffLL * 4 || n * T > 0x7fffffffLL) return fail(TOME_EINVAL, "%s: too many rows", who);
    return TOME_OK;
}

extern "C" int tome_merge_wavg(const void *x, int x_dtype, const void *size, int size_dtype, int64_t n, int64_t T,
                               int64_t C, int64_t r, const int64_t *src_idx, const int64_t *dst_idx,
                               const int64_t *unm_idx, int distill_token, const uint8_t *edge_keep, void *x_out,
                               void *size_out, void *log_size_out, tome_stream_t stream) {
    if (int rc = check_merge_args("tome_merge_wavg", x, n, T, C, r, x_out)) return rc;
    if (!src_idx || !dst_idx || (!unm_idx && (T + 1) / 2 > r) || !size_out)
        return fail(TOME_EINVAL, "tome_merge_wavg: null buffer");
    hipStream_t st = (hipStream_t)stream;
#define WAVG(TX, TS)                                                                                         \
    return launch_merge_rows<TX, TS, OP_WAVG>(x, size, n, T, C, r, src_idx, dst_idx, unm_idx, distill_token, \
                                              edge_keep, x_out, size_out, st, nullptr, nullptr, 0, nullptr, log_size_out)
    if (x_dtype == TOME_F32 && size_dtype == TOME_F32) WAVG(float, float);
    if (x_dtype == TOME_BF16 && size_dtype == TOME_BF16) WAVG(bf16_t, bf16_t);
    if (x_dtype == TOME_BF16 && size_dtype == TOME_F32) WAVG(bf16_t, float);
    if (x_dtype == TOME_F16 && size_dtype == TOME_F16) WAVG(f16_t, f16_t);
    if (x_dtype == TOME_F16 && size_dtype == TOME_F32) WAVG(f16_t, float);
#undef WAVG
    return fail(TOME_EINVAL, "tome_merge_wavg: unsupported dtypes x=%d size=%d", x_dtype, size_dtype);
}

extern "C" int tome_merge_wavg_ln(const void *x, int x_dtype, const void *size, int size_dtype, int64_t n, int64_t T,
                                  int64_t C, int64_t r, const int64_t *src_idx, const int64_t *dst_idx,
                                  const int64_t *unm_idx, int distill_token, const uint8_t *edge_keep,
                                  const void *ln_weight, const void *ln_bias, float eps, const void *addend,
                                  void *x_out, void *y_out, void *size_out, void *log_size_out,
                                  const void *x_out_bias, tome_stream_t stream) {
    if (int rc = check_merge_args("tome_merge_wavg_ln", x, n, T, C, r, x_out)) return rc;
    if (addend && !aligned16(addend)) return fail(TOME_EINVAL, "tome_merge_wavg_ln: addend not 16-byte aligned");
    if (x_out_bias && !aligned16(x_out_bias)) return fail(TOME_EINVAL, "tome_merge_wavg_ln: x_out_bias not 16-byte aligned");
    if (!src_idx || !dst_idx || (!unm_idx && (T + 1) / 2 > r) || !size_out || !y_out || !ln_weight || !ln_bias)
        return fail(TOME_EINVAL, "tome_merge_wavg_ln: null buffer");
    const LnArgs ln{ln_weight, ln_bias, y_out, eps, addend, 0, TokLayout{0, 0, 0, 0, 1}, nullptr, 0, x_out_bias};
    hipStream_t st = (hipStream_t)stream;
#define WAVGLN(TX, TS)                                                                                         \
    return launch_merge_rows<TX, TS, OP_WAVG>(x, size, n, T, C, r, src_idx, dst_idx, unm_idx, distill_token,   \
                                              edge_keep, x_out, size_out, st, nullptr, nullptr, 0, &ln, log_size_out)
    if (x_dtype == TOME_BF16 && size_dtype == TOME_BF16) WAVGLN(bf16_t, bf16_t);
    if (x_dtype == TOME_BF16 && size_dtype == TOME_F32) WAVGLN(bf16_t, float);
    if (x_dtype == TOME_F16 && size_dtype == TOME_F16) WAVGLN(f16_t, f16_t);
    if (x_dtype == TOME_F16 && size_dtype == TOME_F32) WAVGLN(f16_t, float);
#undef WAVGLN
    return fail(TOME_EINVAL, "tome_merge_wavg_ln: 16-bit tokens only (x=%d size=%d)", x_dtype, size_dtype);
}

static int merge_wavg_regrouped_impl(const char *who, const void *x, int x_dtype, const void *size, int size_dtype,
                                     int64_t B, int64_t F, int64_t P, int64_t C, int64_t r, int has_cls,
                                     const int64_t *src_idx, const int64_t *dst_idx, const int64_t *unm_idx,
                                     const uint8_t *edge_keep, void *x_out, void *size_out, void *log_size_out,
                                     const LnArgs *ln, tome_stream_t stream) {
    if (B <= 0 || F <= 0) return fail(TOME_EINVAL, "%s: bad shape", who);
    const int64_t n = B * F;
    if (int rc = check_merge_args(who, x, n, P, C, r, x_out)) return rc;
    if (!src_idx || !dst_idx || (!unm_idx && (P + 1) / 2 > r) || !size_out) return fail(TOME_EINVAL, "%s: null buffer", who);
    const int cls = has_cls ? 1 : 0;
    const TokLayout lin{cls * C, (cls + P * F) * C, C, F * C, (int)F};
    const TokLayout lout{cls * C, (cls + (P - r) * F) * C, C, F * C, (int)F};
    hipStream_t st = (hipStream_t)stream;
#define WAVGR(TX, TS)                                                                                         \
    return launch_merge_rows<TX, TS, OP_WAVG>(x, size, n, P, C, r, src_idx, dst_idx, unm_idx, 0, edge_keep, x_out, \
                                              size_out, st, &lin, &lout, cls ? (int)B : 0, ln, log_size_out)
    if (x_dtype == TOME_F32 && size_dtype == TOME_F32) WAVGR(float, float);
    if (x_dtype == TOME_BF16 && size_dtype == TOME_BF16) WAVGR(bf16_t, bf16_t);
    if (x_dtype == TOME_BF16 && size_dtype == TOME_F32) WAVGR(bf16_t, float);
    if (x_dtype == TOME_F16 && size_dtype == TOME_F16) WAVGR(f16_t, f16_t);
    if (x_dtype == TOME_F16 && size_dtype == TOME_F32) WAVGR(f16_t, float);
#undef WAVGR
    return fail(TOME_EINVAL, "%s: unsupported dtypes x=%d size=%d", who, x_dtype, size_dtype);
}

extern "C" int tome_merge_wavg_regrouped(const void *x, int x_dtype, const void *size, int size_dtype, int64_t B,
                                         int64_t F, int64_t P, int64_t C, int64_t r, int has_cls,
                                         const int64_t *src_idx, const int64_t *dst_idx, const int64_t *unm_idx,
                                         const uint8_t *edge_keep, void *x_out, void *size_out,
                                         void *log_size_out, tome_stream_t stream) {
    return merge_wavg_regrouped_impl("tome_merge_wavg_regrouped", x, x_dtype, size, size_dtype, B, F, P, C, r, has_cls,
                                     src_idx, dst_idx, unm_idx, edge_keep, x_out, size_out, log_size_out, nullptr, stream);
}

extern "C" int tome_merge_wavg_regrouped_ln(const void *x, int x_dtype, const void *size, int size_dtype, int64_t B,
                                            int64_t F, int64_t P, int64_t C, int64_t r, int has_cls,
                                            const int64_t *src_idx, const int64_t *dst_idx,
                                            const int64_t *unm_idx, const uint8_t *edge_keep, const void *ln_weight,
                                            const void *ln_bias, float eps, const void *addend,
                                            int addend_grouped, const void *cls_addend, void *x_out,
                                            void *y_out, void *size_out, void *log_size_out,
                                            const void *x_out_bias, tome_stream_t stream) {
    if (x_out_bias && !aligned16(x_out_bias))
        return fail(TOME_EINVAL, "tome_merge_wavg_regrouped_ln: x_out_bias not 16-byte aligned");
    if (!y_out || !ln_weight || !ln_bias) return fail(TOME_EINVAL, "tome_merge_wavg_regrouped_ln: null buffer");
    if (x_dtype == TOME_F32) return fail(TOME_EINVAL, "tome_merge_wavg_regrouped_ln: 16-bit tokens only");
    if ((addend && !aligned16(addend)) || (cls_addend && !aligned16(cls_addend)))
        return fail(TOME_EINVAL, "tome_merge_wavg_regrouped_ln: addend alignment");
    if (addend_grouped && !addend) return fail(TOME_EINVAL, "tome_merge_wavg_regrouped_ln: addend_grouped without addend");
    LnArgs ln{ln_weight, ln_bias, y_out, eps, addend, 0, TokLayout{0, 0, 0, 0, 1}, nullptr, 0, x_out_bias};
    if (addend_grouped) {  // addend [B*F, has_cls + P, C]: group g's token p at (g*(cls+P) + cls + p)*C
        const int64_t cls = has_cls ? 1 : 0;
        ln.a_own = 1;
        ln.la = TokLayout{cls * C, (cls + P) * C, 0, C, 1};
        ln.cls_addend = cls_addend;
    }
    return merge_wavg_regrouped_impl("tome_merge_wavg_regrouped_ln", x, x_dtype, size, size_dtype, B, F, P, C, r,
                                     has_cls, src_idx, dst_idx, unm_idx, edge_keep, x_out, size_out, log_size_out, &ln, stream);
}

static int add_layernorm_impl(const void *x, const void *addend, int dtype, int64_t rows, int64_t C,
                              const void *ln_weight, const void *ln_bias, float eps, void *x_out, void *y_out,
                              int64_t y_group, tome_stream_t stream);

extern "C" int tome_add_layernorm(const void *x, const void *addend, int dtype, int64_t rows, int64_t C,
                                  const void *ln_weight, const void *ln_bias, float eps, void *x_out, void *y_out,
                                  tome_stream_t stream) {
    return add_layernorm_impl(x, addend, dtype, rows, C, ln_weight, ln_bias, eps, x_out, y_out, 0, stream);
}

extern "C" int tome_add_layernorm_skip_first(const void *x, const void *addend, int dtype, int64_t groups,
                                             int64_t group_rows, int64_t C, const void *ln_weight,
                                             const void *ln_bias, float eps, void *x_out, void *y_out,
                                             tome_stream_t stream) {
    if (groups <= 0 || group_rows < 2 || group_rows > 0x7fffffffLL)
        return fail(TOME_EINVAL, "tome_add_layernorm_skip_first: groups of at least two rows required");
    return add_layernorm_impl(x, addend, dtype, groups * group_rows, C, ln_weight, ln_bias, eps, x_out, y_out, group_rows,
                              stream);
}

static int add_layernorm_impl(const void *x, const void *addend, int dtype, int64_t rows, int64_t C,
                              const void *ln_weight, const void *ln_bias, float eps, void *x_out, void *y_out,
                              int64_t y_group, tome_stream_t stream) {
    // addend == NULL: LayerNorm only (y_out = LN(x)); x_out is then neither read nor written and may be NULL
    if (!x || !ln_weight || !ln_bias || (addend && !x_out) || !y_out || rows <= 0 || C <= 0)
        return fail(TOME_EINVAL, "tome_add_layernorm: bad shape/pointer");
    if (dtype != TOME_BF16 && dtype != TOME_F16) return fail(TOME_EINVAL, "tome_add_layernorm: 16-bit tokens only");
    const int64_t cpr = C / 8;
    if (C % 8 || cpr > 2 * WAVE || !aligned16(x) || !aligned16(addend) || (addend && !aligned16(x_out)) ||
        !aligned16(y_out) || !aligned16(ln_weight) || !aligned16(ln_bias))
        return fail(TOME_EINVAL, "tome_add_layernorm: C %% 8 == 0, C <= 1024 and 16-byte aligned buffers required");
    static const int nit_env = [] {
        const char *e = getenv("TOME_ADD_LN_NIT");  // 3 chunks per lane: 100.6 us vs 105 us with 6 (batch 64)
        int v = e ? atoi(e) : 0;
        return (v == 3 || v == 6) ? v : 3;
    }();
    const int nit = (cpr <= 3 * WAVE) ? nit_env : FAST_NIT;
    int R = (int)((nit * WAVE) / cpr);
    if (R > FAST_MAXR) R = FAST_MAXR;
    const int64_t waves = (rows + R - 1) / R;
    const LnArgs ln{ln_weight, ln_bias, y_out, eps, nullptr, 0, TokLayout{0, 0, 0, 0, 1}, nullptr, (int)y_group, nullptr};
    const dim3 grid((unsigned)((waves + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
#define ADDLN(TX, N)                                                                                              \
    hipLaunchKernelGGL((k_add_ln_rows<TX, N>), grid, dim3(256), 0, st, (const TX *)x, (const TX *)addend, rows, \
                       (int)C, R, (int)cpr, ln, (TX *)x_out)
    if (dtype == TOME_BF16) {
        if (nit == 3) ADDLN(bf16_t, 3); else ADDLN(bf16_t, 6);
    } else {
        if (nit == 3) ADDLN(f16_t, 3); else ADDLN(f16_t, 6);
    }
#undef ADDLN
    return check_launch("k_add_ln_rows");
}

extern "C" int tome_add_layernorm_regrouped(const void *x, const void *addend, int dtype, int64_t B, int64_t F, int64_t P,
                                            int64_t C, const void *ln_weight, const void *ln_bias, float eps,
                                            void *x_out, void *y_out, tome_stream_t stream) {
    if (!x || !addend || !x_out || !y_out || !ln_weight || !ln_bias || B <= 0 || F <= 0 || P <= 0 || C <= 0)
        return fail(TOME_EINVAL, "tome_add_layernorm_regrouped: bad shape/pointer");
    if (dtype != TOME_BF16 && dtype != TOME_F16)
        return fail(TOME_EINVAL, "tome_add_layernorm_regrouped: 16-bit tokens only");
    const int64_t cpr = C / 8;
    if (C % 8 || cpr > 2 * WAVE || !aligned16(x) || !aligned16(addend) || !aligned16(x_out) || !aligned16(y_out) ||
        !aligned16(ln_weight) || !aligned16(ln_bias))
        return fail(TOME_EINVAL, "tome_add_layernorm_regrouped: C %% 8 == 0, C <= 1024 and 16-byte aligned buffers required");
    const int64_t rows = B * (1 + P * F);
    if (rows > 0x7fffffffLL) return fail(TOME_EINVAL, "tome_add_layernorm_regrouped: too many rows");
    const int nit = 3;
    int R = (int)((nit * WAVE) / cpr);
    if (R > FAST_MAXR) R = FAST_MAXR;
    if (R < 1) return fail(TOME_EINVAL, "tome_add_layernorm_regrouped: row too wide");
    const int64_t waves = (rows + R - 1) / R;
    const LnArgs ln{ln_weight, ln_bias, y_out, eps, nullptr, 0, TokLayout{0, 0, 0, 0, 1}, nullptr, 0, nullptr};
    const dim3 grid((unsigned)((waves + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TOME_BF16)
        hipLaunchKernelGGL((k_add_ln_regroup<bf16_t, 3>), grid, dim3(256), 0, st, (const bf16_t *)x, (const bf16_t *)addend,
                           (int)B, (int)F, (int)P, (int)C, R, (int)cpr, ln, (bf16_t *)x_out);
    else
        hipLaunchKernelGGL((k_add_ln_regroup<f16_t, 3>), grid, dim3(256), 0, st, (const f16_t *)x, (const f16_t *)addend,
                           (int)B, (int)F, (int)P, (int)C, R, (int)cpr, ln, (f16_t *)x_out);
    return check_launch("k_add_ln_regroup");
}

template <typename TX>
static int merge_mode_dispatch(int mode, const void *x, int64_t n, int64_t T, int64_t C, int64_t r,
                               const int64_t *src, const int64_t *dst, const int64_t *unm, int distill,
                               const uint8_t *keep, void *out, hipStream_t st) {
    switch (mode) {
    case TOME_SUM: return launch_merge_rows<TX, float, TOME_SUM>(x, nullptr, n, T, C, r, src, dst, unm, distill, keep, out, nullptr, st);
    case TOME_MEAN: return launch_merge_rows<TX, float, TOME_MEAN>(x, nullptr, n, T, C, r, src, dst, unm, distill, keep, out, nullptr, st);
    case TOME_AMAX: return launch_merge_rows<TX, float, TOME_AMAX>(x, nullptr, n, T, C, r, src, dst, unm, distill, keep, out, nullptr, st);
    case TOME_PROD: return launch_merge_rows<TX, float, TOME_PROD>(x, nullptr, n, T, C, r, src, dst, unm, distill, keep, out, nullptr, st);
    case TOME_AMIN: return launch_merge_rows<TX, float, TOME_AMIN>(x, nullptr, n, T, C, r, src, dst, unm, distill, keep, out, nullptr, st);
    case OP_DROP: return launch_merge_rows<TX, float, OP_DROP>(x, nullptr, n, T, C, r, src, dst, unm, distill, keep, out, nullptr, st);
    }
    return fail(TOME_EINVAL, "tome_merge: mode %d", mode);
}

static int merge_dtype_dispatch(const char *who, int dtype, int mode, const void *x, int64_t n, int64_t T,
                                int64_t C, int64_t r, const int64_t *src, const int64_t *dst, const int64_t *unm,
                                int distill, const uint8_t *keep, void *out, hipStream_t st) {
    switch (dtype) {
    case TOME_F32: return merge_mode_dispatch<float>(mode, x, n, T, C, r, src, dst, unm, distill, keep, out, st);
    case TOME_BF16: return merge_mode_dispatch<bf16_t>(mode, x, n, T, C, r, src, dst, unm, distill, keep, out, st);
    case TOME_F16: return merge_mode_dispatch<f16_t>(mode, x, n, T, C, r, src, dst, unm, distill, keep, out, st);
    }
    return fail(TOME_EINVAL, "%s: dtype %d", who, dtype);
}

extern "C" int tome_merge(const void *x, int dtype, int64_t n, int64_t T, int64_t C, int64_t r,
                          const int64_t *src_idx, const int64_t *dst_idx, const int64_t *unm_idx, int distill_token,
                          int mode, const uint8_t *edge_keep, void *out, tome_stream_t stream) {
    if (int rc = check_merge_args("tome_merge", x, n, T, C, r, out)) return rc;
    if (!src_idx || !dst_idx || (!unm_idx && (T + 1) / 2 > r))
        return fail(TOME_EINVAL, "tome_merge: null index buffer");
    if (mode < TOME_SUM || mode > TOME_AMIN) return fail(TOME_EINVAL, "tome_merge: mode %d", mode);
    return merge_dtype_dispatch("tome_merge", dtype, mode, x, n, T, C, r, src_idx, dst_idx, unm_idx, distill_token,
                                edge_keep, out, (hipStream_t)stream);
}

extern "C" int tome_drop(const void *x, int dtype, int64_t n, int64_t T, int64_t C, int64_t r,
                         const int64_t *und_idx, int distill_token, void *out, tome_stream_t stream) {
    if (int rc = check_merge_args("tome_drop", x, n, T, C, r, out)) return rc;
    if (!und_idx && (T + 1) / 2 > r) return fail(TOME_EINVAL, "tome_drop: null index buffer");
    return merge_dtype_dispatch("tome_drop", dtype, OP_DROP, x, n, T, C, r, nullptr, nullptr, und_idx, distill_token,
                                nullptr, out, (hipStream_t)stream);
}

// tome_drop on the interleaved layout of TimeSformer / Motionformer (class token kept aside and copied through)
extern "C" int tome_drop_regrouped(const void *x, int dtype, int64_t B, int64_t F, int64_t P, int64_t C, int64_t r,
                                   int has_cls, const int64_t *und_idx, void *x_out, tome_stream_t stream) {
    if (B <= 0 || F <= 0) return fail(TOME_EINVAL, "tome_drop_regrouped: bad shape");
    const int64_t n = B * F;
    if (int rc = check_merge_args("tome_drop_regrouped", x, n, P, C, r, x_out)) return rc;
    if (!und_idx && (P + 1) / 2 > r) return fail(TOME_EINVAL, "tome_drop_regrouped: null index buffer");
    const int cls = has_cls ? 1 : 0;
    const TokLayout lin{cls * C, (cls + P * F) * C, C, F * C, (int)F};
    const TokLayout lout{cls * C, (cls + (P - r) * F) * C, C, F * C, (int)F};
    hipStream_t st = (hipStream_t)stream;
#define DROPR(TX)                                                                                              \
    return launch_merge_rows<TX, float, OP_DROP>(x, nullptr, n, P, C, r, nullptr, nullptr, und_idx, 0, nullptr, \
                                                 x_out, nullptr, st, &lin, &lout, cls ? (int)B : 0)
    if (dtype == TOME_F32) DROPR(float);
    if (dtype == TOME_BF16) DROPR(bf16_t);
    if (dtype == TOME_F16) DROPR(f16_t);
#undef DROPR
    return fail(TOME_EINVAL, "tome_drop_regrouped: dtype %d", dtype);
}

static int prop_attention_impl(const void *q, const void *k, const void *v, int dtype, int64_t B, int64_t H,
                               int64_t N, int64_t Nk, int64_t D, const int64_t *q_strides,
                               const int64_t *k_strides, const int64_t *v_strides, const float *log_size,
                               int64_t log_size_stride, int bias_skip, float scale, void *out,
                               const int64_t *out_strides, int64_t nseg, const int64_t *seg_strides,
                               tome_stream_t stream);

extern "C" int tome_prop_attention(const void *q, const void *k, const void *v, int dtype, int64_t B, int64_t H,
                                   int64_t N, int64_t Nk, int64_t D, const int64_t *q_strides,
                                   const int64_t *k_strides, const int64_t *v_strides, const float *log_size,
                                   int64_t log_size_stride, int bias_skip, float scale, void *out,
                                   const int64_t *out_strides, tome_stream_t stream) {
    return prop_attention_impl(q, k, v, dtype, B, H, N, Nk, D, q_strides, k_strides, v_strides, log_size,
                               log_size_stride, bias_skip, scale, out, out_strides, 1, nullptr, stream);
}

extern "C" int tome_prop_attention_segments(const void *q, const void *k, const void *v, int dtype, int64_t B,
                                            int64_t H, int64_t N, int64_t Nk, int64_t D, const int64_t *q_strides,
                                            const int64_t *k_strides, const int64_t *v_strides, const float *log_size,
                                            int64_t log_size_stride, float scale, void *out,
                                            const int64_t *out_strides, int64_t nseg, const int64_t *seg_strides,
                                            tome_stream_t stream) {
    if (nseg < 1 || !seg_strides || !out_strides)
        return fail(TOME_EINVAL, "tome_prop_attention_segments: nseg >= 1, segment and out strides required");
    return prop_attention_impl(q, k, v, dtype, B, H, N, Nk, D, q_strides, k_strides, v_strides, log_size,
                               log_size_stride, 0, scale, out, out_strides, nseg, seg_strides, stream);
}

static int prop_attention_impl(const void *q, const void *k, const void *v, int dtype, int64_t B, int64_t H,
                               int64_t N, int64_t Nk, int64_t D, const int64_t *q_strides,
                               const int64_t *k_strides, const int64_t *v_strides, const float *log_size,
                               int64_t log_size_stride, int bias_skip, float scale, void *out,
                               const int64_t *out_strides, int64_t nseg, const int64_t *seg_strides,
                               tome_stream_t stream) {
    if (!q || !k || !v || !out || !q_strides || !k_strides || !v_strides || B <= 0 || H <= 0 || N <= 0 || Nk <= 0)
        return fail(TOME_EINVAL, "tome_prop_attention: bad shape/pointer");
    if (D != ATT_D) return fail(TOME_EINVAL, "tome_prop_attention: head dim %lld (only 64)", (long long)D);
    if (dtype != TOME_BF16 && dtype != TOME_F16) return fail(TOME_EINVAL, "tome_prop_attention: 16-bit q/k/v only");
    if (bias_skip != 0 && bias_skip != 1) return fail(TOME_EINVAL, "tome_prop_attention: bias_skip %d", bias_skip);
    if (bias_skip && N != Nk) return fail(TOME_EINVAL, "tome_prop_attention: bias_skip needs as many keys as queries");
    if (B * H * nseg > 0x7fffffffLL / 64 || N > 0x7fffffffLL / 4 || Nk > 0x7fffffffLL / 4)
        return fail(TOME_EINVAL, "tome_prop_attention: too large");
    const int64_t *ss[3] = {q_strides, k_strides, v_strides};
    const void *pp[3] = {q, k, v};
    for (int i = 0; i < 3; ++i) {
        if (!aligned16(pp[i]) || ss[i][0] % 8 || ss[i][1] % 8 || ss[i][2] % 8 || ss[i][2] < D)
            return fail(TOME_EINVAL, "tome_prop_attention: q/k/v rows must be 16-byte aligned (strides %% 8 == 0)");
    }
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.out = out;
    a.q_sb = q_strides[0]; a.q_sh = q_strides[1]; a.q_sn = q_strides[2];
    a.k_sb = k_strides[0]; a.k_sh = k_strides[1]; a.k_sn = k_strides[2];
    a.v_sb = v_strides[0]; a.v_sh = v_strides[1]; a.v_sn = v_strides[2];
    if (out_strides) {  // {batch, head, token} element strides of out[b, q, h, 0..63]; rows 16-byte aligned
        if (out_strides[0] % 8 || out_strides[1] % 8 || out_strides[2] % 8 || ((uintptr_t)out & 15))
            return fail(TOME_EINVAL, "tome_prop_attention: out rows must be 16-byte aligned");
        a.o_sb = out_strides[0]; a.o_sh = out_strides[1]; a.o_sn = out_strides[2];
    } else {
        if ((uintptr_t)out & 15) return fail(TOME_EINVAL, "tome_prop_attention: out must be 16-byte aligned");
        a.o_sb = N * H * D; a.o_sh = D; a.o_sn = H * D;
    }
    a.log_size = log_size; a.ls_sb = log_size_stride;
    a.B = (int)B; a.H = (int)H; a.N = (int)N; a.Nk = (int)Nk; a.scale = scale; a.bias_skip = bias_skip;
    a.nseg = (int)nseg;
    a.k_seg = a.v_seg = a.o_seg = a.ls_seg = 0;
    if (seg_strides) {  // {k, v, out, log_size} element offsets from one segment to the next
        if (seg_strides[0] % 8 || seg_strides[1] % 8 || seg_strides[2] % 8)
            return fail(TOME_EINVAL, "tome_prop_attention_segments: segment offsets must keep rows 16/8-byte aligned");
        a.k_seg = seg_strides[0]; a.v_seg = seg_strides[1]; a.o_seg = seg_strides[2]; a.ls_seg = seg_strides[3];
    }
    hipStream_t st = (hipStream_t)stream;
    // Short key sequences (TimeSformer's 1 + p <= 197 tokens per frame, Motionformer's <= 196 keys per frame segment):
    // the whole K / V of a (batch, head, segment) resident in LDS, one workgroup per item, no per-tile barrier
    // (tome_attn_resident.h).  TOME_ATTN_RESIDENT=0 keeps the streaming kernels (measurement switch, read per call).
    {
        const char *re = getenv("TOME_ATTN_RESIDENT");
        // (the kernel addresses the tokens of one (batch, head) slice with 32-bit element offsets)
        const bool off32 = N * a.q_sn < (1ll << 31) && Nk * a.k_sn < (1ll << 31) && Nk * a.v_sn < (1ll << 31) &&
                           N * a.o_sn < (1ll << 31);
        if (Nk <= RES_ROWS && off32 && !(re && re[0] == '0')) {
            const int64_t items = (B * H + 7) / 8 * 8 * nseg;
            if (items > 0x7fffffffLL) return fail(TOME_EINVAL, "tome_prop_attention: grid too large");
            const dim3 rgrid((unsigned)items);
            if (dtype == TOME_BF16) {
                if (log_size) hipLaunchKernelGGL((k_resident_attention<bf16_t, true>), rgrid, dim3(512), 0, st, a);
                else hipLaunchKernelGGL((k_resident_attention<bf16_t, false>), rgrid, dim3(512), 0, st, a);
            } else {
                if (log_size) hipLaunchKernelGGL((k_resident_attention<f16_t, true>), rgrid, dim3(512), 0, st, a);
                else hipLaunchKernelGGL((k_resident_attention<f16_t, false>), rgrid, dim3(512), 0, st, a);
            }
            return check_launch("k_resident_attention");
        }
    }
    // queries per workgroup: 256 (eight waves share every staged K/V tile: staging costs 18 % with four) unless the
    // sequence is short.  (Measured: 5, 6 or 7 waves per workgroup, chosen to leave no part-empty last block, are
    // 10-30 % slower per block than eight -- uneven staging passes and SIMD load -- and lose more than they save.)
    // (read per call, not cached: the tests force each workgroup shape on small inputs)
    const int waves_env = [] {
        const char *e = getenv("TOME_ATTN_WAVES");
        int v = e ? atoi(e) : 0;
        return (v == 4 || v == 8) ? v : 0;
    }();
    // (the pipelined plain kernel keeps two waves per SIMD either way: two 4-wave workgroups share a CU.  They lose
    // 3-5 % on long launches -- every tile is staged twice per CU -- and win 7-13 % when the launch is short: fewer
    // than four rounds of 8-wave workgroups over the 256 CUs)
    const int waves = waves_env ? waves_env : ((N > 128 && B * H * nseg * ((N + 255) / 256) >= 1024) ? 8 : 4);
    const int64_t qblocks = (N + 32 * waves - 1) / (32 * waves);
    const int64_t bh8 = (B * H * nseg + 7) / 8 * 8;
    if (bh8 * qblocks > 0x7fffffffLL) return fail(TOME_EINVAL, "tome_prop_attention: grid too large");
    const dim3 grid((unsigned)(bh8 * qblocks));
    // Eight-wave launches with at least two key tiles run as persistent workgroups, one per CU, that keep the K/V
    // pipeline going across query blocks (tome_attn_stream.h); TOME_ATTN_STREAM=0 keeps one workgroup per block
    // (measurement switch, read per call)
    const char *se = getenv("TOME_ATTN_STREAM");
    const int64_t sn_max = 1 << 22;  // (the stream kernel keeps token offsets inside a tile / query block in 32 bits)
    const bool sn_ok = a.q_sn < sn_max && a.k_sn < sn_max && a.v_sn < sn_max && a.o_sn < sn_max;
    // (Round 3: also for launches the rule above gives four waves, as long as a block has work for more than four --
    // 8 x 12 x 1568: 597 -> 650 TFLOP/s, with the per-key bias 476 -> 537; 64 x 12 x 197: 242 -> 280 / 199 -> 240;
    // level at 16 x 12 x 197 and below, where the launch is the cost; the 4-wave persistent form was built and
    // measured too: 17-20 % slower than this one at 197 .. 1568 tokens, not kept)
    const bool stream_ok = Nk > ATT_BN && sn_ok && !(se && se[0] == '0');
    if (stream_ok && (waves == 8 || (!waves_env && N > 128))) {
        const int64_t qblocks = (N + 255) / 256;
        static const int cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
                n = 256;
            return n / 8 * 8;
        }();
        const int nitems = (int)(bh8 * qblocks);
        const dim3 pgrid((unsigned)(nitems < cus ? nitems : cus));
        if (dtype == TOME_BF16) {
            if (log_size) hipLaunchKernelGGL((k_prop_attention_stream<bf16_t, true>), pgrid, dim3(512), 0, st, a, nitems);
            else hipLaunchKernelGGL((k_prop_attention_stream<bf16_t, false>), pgrid, dim3(512), 0, st, a, nitems);
        } else {
            if (log_size) hipLaunchKernelGGL((k_prop_attention_stream<f16_t, true>), pgrid, dim3(512), 0, st, a, nitems);
            else hipLaunchKernelGGL((k_prop_attention_stream<f16_t, false>), pgrid, dim3(512), 0, st, a, nitems);
        }
        return check_launch("k_prop_attention_stream");
    }
#define ATT_LAUNCH(TX, BI)                                                                        \
    if (waves == 8) hipLaunchKernelGGL((k_prop_attention<TX, 8, BI>), grid, dim3(512), 0, st, a); \
    else hipLaunchKernelGGL((k_prop_attention<TX, 4, BI>), grid, dim3(256), 0, st, a);
    if (dtype == TOME_BF16) {
        if (log_size) { ATT_LAUNCH(bf16_t, true) } else { ATT_LAUNCH(bf16_t, false) }
    } else {
        if (log_size) { ATT_LAUNCH(f16_t, true) } else { ATT_LAUNCH(f16_t, false) }
    }
#undef ATT_LAUNCH
    return check_launch("k_prop_attention");
}

#ifdef ATT_DIAG
// diagnostic build: the phase stamps of the last k_prop_attention launch (tools/attn_diag.py)
extern "C" int tome_attn_diag_read(unsigned long long *host, int64_t count) {
    if (count > (int64_t)ATT_DIAG_WGS * 8 * ATT_DIAG_N) count = (int64_t)ATT_DIAG_WGS * 8 * ATT_DIAG_N;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_att_stamps), count * sizeof(unsigned long long)) != hipSuccess)
        return fail(TOME_ELAUNCH, "tome_attn_diag_read: copy failed");
    return TOME_OK;
}
#endif

extern "C" int tome_trajectory_mix(const void *q2, const void *k2, const void *val, int dtype, int64_t B, int64_t S,
                                   int64_t F, int64_t H, int64_t D, int64_t k_row_stride, int64_t v_row_stride,
                                   float scale, void *out, int64_t out_batch_stride, float *tattn,
                                   tome_stream_t stream) {
    if (!q2 || !k2 || !val || !out || B <= 0 || S <= 0 || F <= 0 || H <= 0)
        return fail(TOME_EINVAL, "tome_trajectory_mix: bad shape/pointer");
    if (D != 64 || H > 16 || F > TRAJ_MAXF)
        return fail(TOME_EINVAL, "tome_trajectory_mix: head dim 64, at most 16 heads and %d frames", TRAJ_MAXF);
    if (dtype != TOME_BF16 && dtype != TOME_F16) return fail(TOME_EINVAL, "tome_trajectory_mix: 16-bit tensors only");
    if (k_row_stride % 8 || v_row_stride % 8 || k_row_stride < H * D || v_row_stride < H * D || !aligned16(q2) ||
        !aligned16(k2) || !aligned16(val) || !aligned16(out))
        return fail(TOME_EINVAL, "tome_trajectory_mix: rows must be 16-byte aligned");
    if (out_batch_stride == 0) out_batch_stride = S * H * D;
    if (out_batch_stride < S * H * D || out_batch_stride % 8)
        return fail(TOME_EINVAL, "tome_trajectory_mix: out_batch_stride must be 0 or a multiple of 8 >= S*H*D");
    const int64_t rows = B * S;
    if (rows > 0x7fffffffLL) return fail(TOME_EINVAL, "tome_trajectory_mix: too many tokens");
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TOME_BF16)
        hipLaunchKernelGGL(k_trajectory_mix<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)q2, (const bf16_t *)k2,
                           (const bf16_t *)val, rows, (int)S, (int)F, (int)H, k_row_stride, v_row_stride, scale,
                           (bf16_t *)out, out_batch_stride, tattn);
    else
        hipLaunchKernelGGL(k_trajectory_mix<f16_t>, grid, dim3(256), 0, st, (const f16_t *)q2, (const f16_t *)k2,
                           (const f16_t *)val, rows, (int)S, (int)F, (int)H, k_row_stride, v_row_stride, scale,
                           (f16_t *)out, out_batch_stride, tattn);
    return check_launch("k_trajectory_mix");
}

extern "C" int tome_short_attention(const void *q, const void *k, const void *v, int dtype, int64_t B, int64_t H,
                                    int64_t N, int64_t D, const int64_t *q_strides, const int64_t *k_strides,
                                    const int64_t *v_strides, float scale, void *out, tome_stream_t stream) {
    if (!q || !k || !v || !out || !q_strides || !k_strides || !v_strides || B <= 0 || H <= 0 || N <= 0)
        return fail(TOME_EINVAL, "tome_short_attention: bad shape/pointer");
    if (D != 64 || N > SHORT_MAXN)
        return fail(TOME_EINVAL, "tome_short_attention: head dim 64 and at most %d tokens per sequence", SHORT_MAXN);
    if (dtype != TOME_BF16 && dtype != TOME_F16) return fail(TOME_EINVAL, "tome_short_attention: 16-bit tensors only");
    // {batch, head, token} element strides; the heads of a token lie side by side (head stride 64), rows 16-byte aligned
    const int64_t *strides[3] = {q_strides, k_strides, v_strides};
    for (int i = 0; i < 3; ++i)
        if (strides[i][1] != 64 || strides[i][0] % 8 || strides[i][2] % 8)
            return fail(TOME_EINVAL, "tome_short_attention: head stride must be 64, batch / token strides multiples of 8");
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out))
        return fail(TOME_EINVAL, "tome_short_attention: rows must be 16-byte aligned");
    const int64_t units = B * H;  // (sequence, head) pairs, eight lanes each
    const int64_t blocks = (units + 31) / 32;
    if (blocks > 0x7fffffffLL) return fail(TOME_EINVAL, "tome_short_attention: too many sequences");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TOME_BF16)
        hipLaunchKernelGGL(k_short_attention<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t *)q,
                           (const bf16_t *)k, (const bf16_t *)v, q_strides[0], q_strides[2], k_strides[0], k_strides[2],
                           v_strides[0], v_strides[2], units, (int)H, (int)N, scale, (bf16_t *)out);
    else
        hipLaunchKernelGGL(k_short_attention<f16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const f16_t *)q,
                           (const f16_t *)k, (const f16_t *)v, q_strides[0], q_strides[2], k_strides[0], k_strides[2],
                           v_strides[0], v_strides[2], units, (int)H, (int)N, scale, (f16_t *)out);
    return check_launch("k_short_attention");
}

extern "C" int tome_gelu_erf(const void *x, int dtype, int64_t elements, void *y, tome_stream_t stream) {
    if (!x || !y || elements <= 0) return fail(TOME_EINVAL, "tome_gelu_erf: bad shape/pointer");
    if (dtype != TOME_BF16 && dtype != TOME_F16) return fail(TOME_EINVAL, "tome_gelu_erf: 16-bit tensors only");
    if (elements % 8 || !aligned16(x) || !aligned16(y))
        return fail(TOME_EINVAL, "tome_gelu_erf: a multiple of 8 elements in 16-byte aligned buffers required");
    const int64_t chunks = elements / 8;
    const int64_t blocks = (chunks + 1023) / 1024;  // 256 threads x 4 chunks
    if (blocks > 0x7fffffffLL) return fail(TOME_EINVAL, "tome_gelu_erf: too large");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TOME_BF16)
        hipLaunchKernelGGL(k_gelu_erf<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t *)x, (bf16_t *)y, chunks);
    else
        hipLaunchKernelGGL(k_gelu_erf<f16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const f16_t *)x, (f16_t *)y, chunks);
    return check_launch("k_gelu_erf");
}

extern "C" int tome_tubelet_rows(const void *x, int elem_bytes, int64_t B, int64_t C, int64_t T, int64_t H, int64_t W,
                                 const int64_t *x_strides, int64_t kt, int64_t kh, int64_t kw, void *rows,
                                 tome_stream_t stream) {
    if (!x || !rows || !x_strides || B <= 0 || C <= 0 || T <= 0 || H <= 0 || W <= 0 || kt <= 0 || kh <= 0 || kw <= 0)
        return fail(TOME_EINVAL, "tome_tubelet_rows: bad shape/pointer");
    if (elem_bytes != 2 && elem_bytes != 4) return fail(TOME_EINVAL, "tome_tubelet_rows: 2- or 4-byte elements only");
    if (T % kt || H % kh || W % kw) return fail(TOME_EINVAL, "tome_tubelet_rows: the clip must be whole tubelets");
    if ((kw * elem_bytes) % 16 || !aligned16(x) || !aligned16(rows))
        return fail(TOME_EINVAL, "tome_tubelet_rows: runs of kw elements must be whole 16-byte chunks in aligned buffers");
    for (int i = 0; i < 4; ++i)
        if (x_strides[i] < 0 || (x_strides[i] * elem_bytes) % 16)
            return fail(TOME_EINVAL, "tome_tubelet_rows: {b, c, t, h} strides must be non-negative multiples of 16 bytes");
    TubeArgs a;
    a.sb = x_strides[0]; a.sc = x_strides[1]; a.st = x_strides[2]; a.sh = x_strides[3];
    a.nt = (int)(T / kt); a.nh = (int)(H / kh); a.nw = (int)(W / kw);
    a.kt = (int)kt; a.kh = (int)kh;
    a.cpr = (int)(kw * elem_bytes / 16);
    const int64_t chunks = C * kt * kh * a.cpr;
    if (chunks > 0x7fffffffLL || T / kt > 0x7fffffffLL || H / kh > 0x7fffffffLL || W / kw > 0x7fffffffLL)
        return fail(TOME_EINVAL, "tome_tubelet_rows: too large");
    a.chunks = (int)chunks;
    a.items = B * a.nt * a.nh * chunks;
    const int64_t blocks = (a.items + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(TOME_EINVAL, "tome_tubelet_rows: too large");
    hipStream_t st = (hipStream_t)stream;
    if (elem_bytes == 2)
        hipLaunchKernelGGL(k_tubelet_rows<2>, dim3((unsigned)blocks), dim3(256), 0, st, (const uint8_t *)x, (uint8_t *)rows, a);
    else
        hipLaunchKernelGGL(k_tubelet_rows<4>, dim3((unsigned)blocks), dim3(256), 0, st, (const uint8_t *)x, (uint8_t *)rows, a);
    return check_launch("k_tubelet_rows");
}

extern "C" int tome_row_map(int64_t n, int64_t T, int64_t r, int distill_token, const int64_t *src_idx,
                            const int64_t *dst_idx, const int64_t *unm_idx, int32_t *row_map, tome_stream_t stream) {
    const int64_t T1 = (T + 1) / 2;
    if (n <= 0 || T <= 0 || r <= 0 || r > T1 || !row_map || !src_idx || !dst_idx || (!unm_idx && T1 > r))
        return fail(TOME_EINVAL, "tome_row_map: bad shape/pointer");
    if (n * T1 > 0x7fffffffLL) return fail(TOME_EINVAL, "tome_row_map: too large");
    hipLaunchKernelGGL(k_row_map, dim3((unsigned)((n * T1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)n,
                       (int)T1, (int)r, distill_token, src_idx, dst_idx, unm_idx, row_map);
    return check_launch("k_row_map");
}

extern "C" int tome_source_init(int64_t n, int64_t T, int64_t r, int distill_token, int drop, const int32_t *row_map,
                                float *source_out, tome_stream_t stream) {
    const int64_t T1 = (T + 1) / 2;
    if (n <= 0 || T <= 0 || r <= 0 || r > T1 || !row_map || !source_out)
        return fail(TOME_EINVAL, "tome_source_init: bad shape/pointer");
    if (n * (T - r) > 0x7fffffffLL) return fail(TOME_EINVAL, "tome_source_init: too many rows");
    hipLaunchKernelGGL(k_source_init, dim3((unsigned)(n * (T - r))), dim3(256), 0, (hipStream_t)stream, (int)n, (int)T,
                       (int)r, distill_token, drop ? 1 : 0, row_map, source_out);
    return check_launch("k_source_init");
}

template <typename TX>
static int launch_unmerge(const void *x, int64_t n, int64_t T, int64_t C, int64_t r, const int64_t *src,
                          const int64_t *dst, const int64_t *unm, void *out, hipStream_t st) {
    constexpr int VEC = 16 / sizeof(TX);
    const int64_t rows = n * (T - r);
    const unsigned nb = (unsigned)((rows + 3) / 4);
    if ((C % VEC == 0) && aligned16(x) && aligned16(out))
        hipLaunchKernelGGL((k_unmerge_rows<TX, VEC>), dim3(nb), dim3(256), 0, st, (const TX *)x, (int)n, (int)T,
                           (int)C, (int)r, src, dst, unm, (TX *)out);
    else
        hipLaunchKernelGGL((k_unmerge_rows<TX, 1>), dim3(nb), dim3(256), 0, st, (const TX *)x, (int)n, (int)T, (int)C,
                           (int)r, src, dst, unm, (TX *)out);
    return check_launch("k_unmerge_rows");
}

extern "C" int tome_unmerge(const void *x, int dtype, int64_t n, int64_t T, int64_t C, int64_t r,
                            const int64_t *src_idx, const int64_t *dst_idx, const int64_t *unm_idx, void *out,
                            tome_stream_t stream) {
    if (int rc = check_merge_args("tome_unmerge", x, n, T, C, r, out)) return rc;
    if (!src_idx || !dst_idx || (!unm_idx && (T + 1) / 2 > r))
        return fail(TOME_EINVAL, "tome_unmerge: null index buffer");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case TOME_F32: return launch_unmerge<float>(x, n, T, C, r, src_idx, dst_idx, unm_idx, out, st);
    case TOME_BF16: return launch_unmerge<bf16_t>(x, n, T, C, r, src_idx, dst_idx, unm_idx, out, st);
    case TOME_F16: return launch_unmerge<f16_t>(x, n, T, C, r, src_idx, dst_idx, unm_idx, out, st);
    }
    return fail(TOME_EINVAL, "tome_unmerge: dtype %d", dtype);
}
