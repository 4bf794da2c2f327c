// tome_match.h -- part of the single translation unit csrc/tome_kernels.hip (matching: unit vectors, MFMA similarity + row max, ranking).
#pragma once
// ------------------------------------------------------------------------------------------------
// Workspace layout of the unit vectors ("fragment-major tiles").  Both sets (A = even tokens, B = odd
// tokens) are cut into tiles of 32 rows; a tile is stored exactly as the 64 lanes of
// v_mfma_f32_32x32x2_f32 consume it, so every operand fetch is one fully coalesced 1-KiB
// global_load_dwordx4 per 4 k-pairs:
//     float4 index inside a group = ((tile * nchunk + c) * 8 + q) * 64 + lane
//     lane = (row & 31) + 32*h holds unit[row][k = 2*s + h] for the 4 pairs s = 32*c + 4*q + {0,1,2,3}
// (lane half h supplies k = 2s+h at MFMA step s).  Dp = D rounded up to 64 (nchunk = Dp/64), the
// padding channels are written as zeros; rows past the end of a set are never written nor used.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t frag_index(int tile, int nchunk, int c, int q, int lane) {
    return (((int64_t)tile * nchunk + c) * 8 + q) * 64 + lane;
}

template <typename T> struct Load8;  // 8 consecutive channels of one token -> fp32
template <> struct Load8<float> {
    static __device__ __forceinline__ void run(const float *p, float (&v)[8]) {
        f32x4 a = *reinterpret_cast<const f32x4 *>(p), b = *reinterpret_cast<const f32x4 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};
template <> struct Load8<bf16_t> {
    static __device__ __forceinline__ void run(const bf16_t *p, float (&v)[8]) { load_pack<bf16_t, 8>(p, v); }
};
template <> struct Load8<f16_t> {
    static __device__ __forceinline__ void run(const f16_t *p, float (&v)[8]) { load_pack<f16_t, 8>(p, v); }
};

// ------------------------------------------------------------------------------------------------
// k_unit_rows: merge.py:51-52.  Eight lanes per token, 16-byte (bf16/fp16) or 2x16-byte (fp32) loads;
// lane b of a token owns the channel blocks b, b+8, ... (8 channels each).  Squared norm in the
// contract's order: fma chain inside a block, blocks added in ascending order (the partials travel
// between the 8 lanes by shuffles).  Each lane then divides its channels and writes two float4 per
// block: the even channels to lane slot (row&31), the odd ones to slot (row&31)+32 of the tile.
// NCH = Dp/64 is a template parameter so that the per-lane partials stay in registers.
// ------------------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ __launch_bounds__(256) void k_unit_rows(const T *__restrict__ metric, int64_t stride_n,
                                                   int64_t stride_t, int n, int T_, int D,
                                                   float *__restrict__ unitA, float *__restrict__ unitB,
                                                   int64_t groupA_f4, int64_t groupB_f4,
                                                   uint8_t *__restrict__ badA, uint8_t *__restrict__ badB) {
    const int lane = threadIdx.x & 63;
    const int b8 = lane & 7;
    const int64_t tok = ((int64_t)blockIdx.x * (blockDim.x >> 3)) + (threadIdx.x >> 3);
    const int64_t ntok = (int64_t)n * T_;
    const bool live = tok < ntok;
    const int64_t tk = live ? tok : ntok - 1;
    const int g = (int)(tk / T_);
    const int t = (int)(tk - (int64_t)g * T_);
    const T *row = metric + (int64_t)g * stride_n + (int64_t)t * stride_t;
    const int nblk = D >> 3;  // D % 8 == 0 on this path

    float v[NCH][8];
    float part[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int b = b8 + 8 * it;
        part[it] = 0.0f;
        if (b < nblk) {
            Load8<T>::run(row + 8 * b, v[it]);
#pragma unroll
            for (int e = 0; e < 8; ++e) part[it] = __fmaf_rn(v[it][e], v[it][e], part[it]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[it][e] = 0.0f;
        }
    }
    float ss = 0.0f;
    const int base = lane & ~7;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const float p = __shfl(part[it], base + l);
            if (l + 8 * it < nblk) ss = __fadd_rn(ss, p);
        }
    }
    const float nr = __builtin_sqrtf(ss);
    const int rowi = t >> 1;
    f32x4 *dst = reinterpret_cast<f32x4 *>((t & 1) ? unitB : unitA) + (int64_t)g * ((t & 1) ? groupB_f4 : groupA_f4);
    const int tile = rowi >> 5, slot = rowi & 31;
    bool nan_here = false;  // a zero / inf / NaN token has NaN unit channels (merge.py:51 has no epsilon)
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int b = b8 + 8 * it;
        f32x4 ev, od;
        if (b < nblk) {
            ev.x = __fdiv_rn(v[it][0], nr); od.x = __fdiv_rn(v[it][1], nr);
            ev.y = __fdiv_rn(v[it][2], nr); od.y = __fdiv_rn(v[it][3], nr);
            ev.z = __fdiv_rn(v[it][4], nr); od.z = __fdiv_rn(v[it][5], nr);
            ev.w = __fdiv_rn(v[it][6], nr); od.w = __fdiv_rn(v[it][7], nr);
            nan_here = nan_here || (ev.x != ev.x) || (ev.y != ev.y) || (ev.z != ev.z) || (ev.w != ev.w) ||
                       (od.x != od.x) || (od.y != od.y) || (od.z != od.z) || (od.w != od.w);
        } else {
            ev.x = ev.y = ev.z = ev.w = 0.0f;
            od = ev;
        }
        // block b -> pairs s = 4b..4b+3 -> chunk c = b/8 = it, q = b%8 = b8
        if (live) {
            const int64_t f = frag_index(tile, NCH, it, b8, slot);
            dst[f] = ev;
            dst[f + 32] = od;
        }
    }
    // one flag per token: does its unit vector hold a NaN (then every score it takes part in is NaN)
    const unsigned long long nan_mask = __ballot(nan_here);
    if (live && b8 == 0) {
        const uint8_t flag = ((nan_mask >> (lane & ~7)) & 0xFFull) ? 1 : 0;
        if (t & 1) badB[(int64_t)g * (T_ >> 1) + rowi] = flag;
        else badA[(int64_t)g * ((T_ + 1) >> 1) + rowi] = flag;
    }
}

// k_unit_rows_heads: the metric producer fused in (videomae.py:72-73 `metric = k.mean(1)`, timesformer.py:83,
// vivit.py:123-124): reads the per-head keys [n,H,T,64] straight from the attention's qkv buffer (any strides
// with unit channel stride), averages the heads exactly as torch does on CPU -- fp32 sum in head order, one
// division by H, one rounding to the keys' dtype -- and continues as k_unit_rows.  D = 64 only (one chunk).
// Groups may be interleaved inside a clip's token sequence (Motionformer, motionformer.py:143-144:
// '(b h) (s f) d -> (b f) h s d'): group g = outer * inner + f lives at outer * stride_n + f * stride_inner and its
// token s at s * stride_t (= inner token rows), so the regrouped keys are read in place as well (inner = 1: plain).
template <typename T>
__global__ __launch_bounds__(256) void k_unit_rows_heads(const T *__restrict__ keys, int64_t stride_n, int inner,
                                                         int64_t stride_inner,
                                                         int64_t stride_h, int64_t stride_t, int n, int H, int T_,
                                                         float *__restrict__ unitA, float *__restrict__ unitB,
                                                         int64_t groupA_f4, int64_t groupB_f4,
                                                         uint8_t *__restrict__ badA, uint8_t *__restrict__ badB) {
    const int lane = threadIdx.x & 63;
    const int b8 = lane & 7;
    const int64_t tok = ((int64_t)blockIdx.x * (blockDim.x >> 3)) + (threadIdx.x >> 3);
    const int64_t ntok = (int64_t)n * T_;
    const bool live = tok < ntok;
    const int64_t tk = live ? tok : ntok - 1;
    // (32-bit division whenever the token count allows it -- the 64-bit one is ~100 vector instructions per lane --
    // and none at all for the group decomposition of the plain layout)
    int g, t;
    if (ntok <= 0x7fffffffLL) {
        g = (int)((uint32_t)tk / (uint32_t)T_);
        t = (int)((uint32_t)tk - (uint32_t)g * (uint32_t)T_);
    } else {
        g = (int)(tk / T_);
        t = (int)(tk - (int64_t)g * T_);
    }
    const T *row = keys + (int64_t)t * stride_t + 8 * b8;
    if (inner == 1) row += (int64_t)g * stride_n;
    else row += (int64_t)(g / inner) * stride_n + (int64_t)(g % inner) * stride_inner;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll 12
    for (int h = 0; h < H; ++h) {
        float kv[8];
        Load8<T>::run(row + (int64_t)h * stride_h, kv);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = __fadd_rn(acc[e], kv[e]);
    }
    const float fh = (float)H;
    float v[8];
    float part = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        v[e] = to_f32(from_f32<T>(__fdiv_rn(acc[e], fh)));  // k.mean(1) in the keys' dtype
        part = __fmaf_rn(v[e], v[e], part);
    }
    float ss = 0.0f;
    const int base = lane & ~7;
#pragma unroll
    for (int l = 0; l < 8; ++l) ss = __fadd_rn(ss, __shfl(part, base + l));
    const float nr = __builtin_sqrtf(ss);
    const int rowi = t >> 1;
    f32x4 *dst = reinterpret_cast<f32x4 *>((t & 1) ? unitB : unitA) + (int64_t)g * ((t & 1) ? groupB_f4 : groupA_f4);
    f32x4 ev, od;
    ev.x = __fdiv_rn(v[0], nr); od.x = __fdiv_rn(v[1], nr);
    ev.y = __fdiv_rn(v[2], nr); od.y = __fdiv_rn(v[3], nr);
    ev.z = __fdiv_rn(v[4], nr); od.z = __fdiv_rn(v[5], nr);
    ev.w = __fdiv_rn(v[6], nr); od.w = __fdiv_rn(v[7], nr);
    const bool nan_here = (ev.x != ev.x) || (ev.y != ev.y) || (ev.z != ev.z) || (ev.w != ev.w) || (od.x != od.x) ||
                          (od.y != od.y) || (od.z != od.z) || (od.w != od.w);
    if (live) {
        const int64_t f = frag_index(rowi >> 5, 1, 0, b8, rowi & 31);
        dst[f] = ev;
        dst[f + 32] = od;
    }
    const unsigned long long nan_mask = __ballot(nan_here);
    if (live && b8 == 0) {
        const uint8_t flag = ((nan_mask >> (lane & ~7)) & 0xFFull) ? 1 : 0;
        if (t & 1) badB[(int64_t)g * (T_ >> 1) + rowi] = flag;
        else badA[(int64_t)g * ((T_ + 1) >> 1) + rowi] = flag;
    }
}

// Any D (also D % 8 != 0, unaligned rows): one thread per token, scalar accesses, same arithmetic order.
template <typename T>
__global__ __launch_bounds__(256) void k_unit_rows_generic(const T *__restrict__ metric, int64_t stride_n,
                                                           int64_t stride_t, int n, int T_, int D, int Dp,
                                                           float *__restrict__ unitA, float *__restrict__ unitB,
                                                           int64_t groupA_f4, int64_t groupB_f4,
                                                           uint8_t *__restrict__ badA, uint8_t *__restrict__ badB) {
    const int64_t tok = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tok >= (int64_t)n * T_) return;
    const int g = (int)(tok / T_);
    const int t = (int)(tok - (int64_t)g * T_);
    const T *row = metric + (int64_t)g * stride_n + (int64_t)t * stride_t;
    float ss = 0.0f;
    for (int k0 = 0; k0 < D; k0 += 8) {
        float part = 0.0f;
        for (int k = k0; k < D && k < k0 + 8; ++k) {
            const float v = to_f32(row[k]);
            part = __fmaf_rn(v, v, part);
        }
        ss = __fadd_rn(ss, part);
    }
    const float nr = __builtin_sqrtf(ss);
    const int rowi = t >> 1, tile = rowi >> 5, slot = rowi & 31, nchunk = Dp >> 6;
    float *dst = ((t & 1) ? unitB : unitA) + 4 * (int64_t)g * ((t & 1) ? groupB_f4 : groupA_f4);
    bool nan_here = false;
    for (int k = 0; k < Dp; ++k) {
        const float u = (k < D) ? __fdiv_rn(to_f32(row[k]), nr) : 0.0f;
        nan_here = nan_here || (u != u);
        const int s = k >> 1, h = k & 1;
        const int64_t f = frag_index(tile, nchunk, s >> 5, (s & 31) >> 2, slot + 32 * h);
        dst[4 * f + (s & 3)] = u;
    }
    if (t & 1) badB[(int64_t)g * (T_ >> 1) + rowi] = nan_here ? 1 : 0;
    else badA[(int64_t)g * ((T_ + 1) >> 1) + rowi] = nan_here ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// k_scores_rowmax: merge.py:53,59-64 without the score matrix, without LDS.
//   One WAVE per work item (group, A tile of 32 rows, j-part): single-wave workgroups, so the
//   dispatcher balances thousands of small items over the 1024 SIMDs and no barrier exists.  The
//   wave keeps its 32 A rows in registers (MFMA "B" operand) and streams its share of the B tiles
//   straight from L2 (fragment-major layout: 8 coalesced 1-KiB loads per 32 rows x 64 channels),
//   one tile ahead of the MFMAs, ping-ponging two register buffers.
//   S^T tile = Bhat_tile (MFMA "A" operand) x Ahat^T: accumulator register v of lane l holds
//   S[i = l&31][j = 32*jt + (v&3) + 8*(v>>2) + 4*(l>>5)], so the max over j is a per-lane running
//   max, merged across the two lane halves at the end.  Each j-part writes its (max, first argmax)
//   to part_max/part_idx [n][WJ][T1]; k_rank_select folds the parts in ascending j order.
//   v_mfma_f32_32x32x2_f32 adds k = 2s then k = 2s+1 to the accumulator: the contract's fma chain.
// ------------------------------------------------------------------------------------------------
#define TILE_ROWS 32
#define MAX_WJ 8
#ifndef SCORES_BBUF
#define SCORES_BBUF 4  // float4 of the B stream in flight per lane (8 = a whole step ahead, 4 = half a step)
#endif

struct RowBest {
    float best;
    int idx;
};

__device__ __forceinline__ void fold_tile(const f32x16 &acc, RowBest &rb, int jt, int h, int T2, int distill_token) {
    const int jbase = jt * TILE_ROWS + 4 * h;
    const bool edge = (jt == 0 && distill_token) || ((jt + 1) * TILE_ROWS > T2);
    if (!edge) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float sc = acc[v];
            const bool up = sc > rb.best;
            rb.best = up ? sc : rb.best;
            rb.idx = up ? jbase + (v & 3) + 8 * (v >> 2) : rb.idx;
        }
    } else {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int j = jbase + (v & 3) + 8 * (v >> 2);
            const float sc = acc[v];
            const bool up = (j < T2) && !(distill_token && j == 0) && (sc > rb.best);
            rb.best = up ? sc : rb.best;
            rb.idx = up ? j : rb.idx;
        }
    }
}

#ifdef TOME_DIAG_CLOCK
// Diagnostic build only (tools/diag_clock.py; never the shipped library): every wave stamps the shader clock
// (s_memtime) and the 100 MHz wall counter (s_memrealtime) around its tile loop; the stamps go to a buffer of
// their own that nothing else reads (MI355X_MICROARCH.md, DVFS give-back item 6).
#define TOME_DIAG_SLOTS 65536
__device__ unsigned long long g_diag_stamps[TOME_DIAG_SLOTS * 2];
#endif

#ifndef SCORES_WAVES
#define SCORES_WAVES 5  // waves per SIMD the register allocation aims at.  Measured at batch 128 (12 layers of VideoMAE-B,
                        // profiles/r02_scores_occupancy.txt): 8 float4 / 4 waves (106 VGPRs) 99.0 TF/s; 4 float4 / 5 waves
                        // (94 VGPRs) 100.4; 4 float4 / 6 waves (80 VGPRs, 60 B of scratch) 77.3 -- occupancy is not the lever
#endif
// (the multi-chunk form -- D > 64: the "concat" metric -- reloads its A fragments per chunk and spilled 12 bytes per
// lane at five waves per SIMD; it is given the registers of four)
template <bool ONE_CHUNK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ONE_CHUNK ? SCORES_WAVES : 4, 8))) void k_scores_rowmax(const f32x4 *__restrict__ unitA,
                                                      const f32x4 *__restrict__ unitB, int n, int T1, int T2,
                                                      int nchunk, int ntA, int ntB, int WJ, int64_t groupA_f4,
                                                      int64_t groupB_f4, int distill_token,
                                                      float *__restrict__ part_max, int *__restrict__ part_idx,
                                                      const uint8_t *__restrict__ tile_flag) {
    // XCD-aware block -> (group, tile, part) map: blocks b and b+8 share an XCD (round-robin dispatch),
    // so all work items of one group -- which stream the same B tiles -- get ids congruent mod 8 and
    // find those tiles in their XCD's L2.  Placement only affects speed.
    const int L = blockIdx.x;
    const int xcd = L & 7, qq = L >> 3;
    const int per_group = ntA * WJ;
    const int g = (qq / per_group) * 8 + xcd;
    if (g >= n) return;
    const int item = qq % per_group;
    const int ti = item / WJ, part = item % WJ;
    // behind the candidate filter (tome_match_filter.h) only the tiles it flagged are computed here
    if (tile_flag && !tile_flag[(int64_t)g * ntA + ti]) return;

    const int lane = threadIdx.x;
    const int col = lane & 31, h = lane >> 5;
    const int i = ti * TILE_ROWS + col;
    // this wave's B tiles: [jt0, jt1), balanced split of ntB over WJ parts
    const int jt0 = (int)(((int64_t)ntB * part) / WJ), jt1 = (int)(((int64_t)ntB * (part + 1)) / WJ);

    const f32x4 *atile = unitA + (int64_t)g * groupA_f4 + frag_index(ti, nchunk, 0, 0, lane);
    const f32x4 *bstream = unitB + (int64_t)g * groupB_f4 + frag_index(jt0, nchunk, 0, 0, lane);

    RowBest rb = {-INFINITY, 0};
    const int nstep = (jt1 - jt0) * nchunk;  // consecutive (tile, chunk) blocks of 512 float4
#ifdef TOME_DIAG_CLOCK
    const unsigned long long diag_c0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (nstep > 0) {
        // B operands: SCORES_BBUF float4 in flight (8: a whole step ahead; 4: half a step ahead, 16 registers
        // fewer -- the form that reaches six waves per SIMD).  Each float4 is replaced right after its last use.
        constexpr int NB = SCORES_BBUF;
        f32x4 af[8], bt[NB];
#pragma unroll
        for (int q = 0; q < 8; ++q) af[q] = atile[q * 64];
#pragma unroll
        for (int q = 0; q < NB; ++q) bt[q] = bstream[q * 64];
        f32x16 acc;
        int jt = jt0, c = 0;
        for (int step = 0; step < nstep; ++step) {
            // the last step re-reads its own block instead of running past the stream
            const int nx = step + 1 < nstep ? step + 1 : step;
            const f32x4 *cb = bstream + (int64_t)step * 512, *nb = bstream + (int64_t)nx * 512;
            const int cn = (c + 1 == nchunk) ? 0 : c + 1;
            if (c == 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 b = bt[q % NB], a = af[q];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc, 0, 0, 0);
                // element q + NB of the stream: this step's second half (NB = 4, q < 4) or the next step's
                bt[q % NB] = (q + NB < 8) ? cb[(q + NB) * 64] : nb[(q + NB - 8) * 64];
                if (!ONE_CHUNK) af[q] = atile[(cn * 8 + q) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c == nchunk - 1) fold_tile(acc, rb, jt, h, T2, distill_token);
            c = cn;
            jt += (cn == 0);
        }
    }
#ifdef TOME_DIAG_CLOCK
    if (lane == 0 && L < TOME_DIAG_SLOTS) {
        g_diag_stamps[2 * L] = __builtin_amdgcn_s_memtime() - diag_c0;
        g_diag_stamps[2 * L + 1] = __builtin_amdgcn_s_memrealtime() - diag_r0;
    }
#endif
    // the two lane halves hold the same A row, disjoint B rows: keep the larger, first index on ties
    {
        const float ob = __shfl_xor(rb.best, 32);
        const int oi = __shfl_xor(rb.idx, 32);
        if (ob > rb.best || (ob == rb.best && oi < rb.idx)) {
            rb.best = ob;
            rb.idx = oi;
        }
    }
    if (h == 0 && i < T1) {
        const int64_t o = ((int64_t)g * WJ + part) * T1 + i;
        part_max[o] = rb.best;
        part_idx[o] = rb.idx;
    }
}

// Row max / first argmax of caller-provided scores (random_merge / random_drop): one wave per row,
// NaN wins and the first NaN keeps the row, like torch.max on CPU.
__global__ __launch_bounds__(256) void k_rowmax_given(const float *__restrict__ scores, int n, int T1,
                                                      int T2, int class_token, int distill_token,
                                                      float *__restrict__ node_max,
                                                      int *__restrict__ node_idx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= (int64_t)n * T1) return;
    const int i = (int)(row % T1);
    const float *s = scores + row * T2;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    bool bnan = false;
    for (int j = lane; j < T2; j += WAVE) {
        float v = s[j];
        if (distill_token && j == 0) v = -INFINITY;
        bool vnan = v != v;
        if (bidx == 0x7fffffff || (!bnan && (vnan || v > best))) {
            best = v;
            bidx = j;
            bnan = vnan;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ob = __shfl_xor(best, off);
        int oi = __shfl_xor(bidx, off);
        bool on = ob != ob;
        bool take;
        if (oi == 0x7fffffff) take = false;
        else if (bidx == 0x7fffffff) take = true;
        else if (bnan || on) take = on && (!bnan || oi < bidx);
        else take = (ob > best) || (ob == best && oi < bidx);
        if (take) {
            best = ob;
            bidx = oi;
            bnan = on;
        }
    }
    if (class_token && i == 0) {
        best = -INFINITY;
        bidx = 0;
    }
    if (lane == 0) {
        node_max[row] = best;
        node_idx[row] = bidx;
    }
}

// ------------------------------------------------------------------------------------------------
// k_rank_select: merge.py:65-69.  rank(i) = #{j : key_j before key_i}, keys descending, NaN first,
// -0 == +0, equal keys in ascending row order.  The order is made total by a 64-bit key
// (sortable score << 32 | ~row); a workgroup ranks 64 rows, 4 lanes per row each counting a quarter of
// the keys held in LDS; the rank IS the position in edge_idx, so src/dst/unm are written directly.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sort_key(float f) {
    if (f != f) return 0xFFFFFFFFu;
    f = f + 0.0f;  // -0 -> +0
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}


__global__ __launch_bounds__(256) void k_rank_select(const float *__restrict__ part_max,
                                                     const int *__restrict__ part_idx, int nparts, int n, int T1,
                                                     int T2, const uint8_t *__restrict__ badA,
                                                     const uint8_t *__restrict__ badB, int r, int class_token,
                                                     int distill_token,
                                                     int64_t *__restrict__ src_idx,
                                                     int64_t *__restrict__ dst_idx,
                                                     int64_t *__restrict__ unm_idx, float *__restrict__ node_max,
                                                     int *__restrict__ rank_out, int *__restrict__ row_map) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int g = blockIdx.y;
    const int quarter = (((T1 + 3) >> 2) + 1) & ~1;  // keys per lane, even (two keys per 16-byte read)
    const int T1p = quarter * 4;
    const float *pm = part_max + (int64_t)g * nparts * T1;
    const int *pi = part_idx + (int64_t)g * nparts * T1;
    const int i = blockIdx.x * 64 + (threadIdx.x >> 2);
    // NaN semantics of torch.max (merge.py:64): a NaN score wins and the FIRST NaN column keeps the row.
    // NaN scores come only from tokens whose unit vector is NaN (flags from k_unit_rows); the MFMA pass
    // ignored them (x > NaN is false), so they are put back here.
    __shared__ int s_first_bad;
    const int jmin = distill_token ? 1 : 0;  // merge.py:61-62: column 0 is -inf when protected
    if (threadIdx.x == 0) s_first_bad = 0x7fffffff;
    __syncthreads();
    if (badB) {
        int fb = 0x7fffffff;
        for (int j = jmin + (int)threadIdx.x; j < T2; j += blockDim.x)
            if (badB[(int64_t)g * T2 + j]) { fb = j; break; }
        if (fb != 0x7fffffff) atomicMin(&s_first_bad, fb);
    }
    __syncthreads();
    const int first_bad = s_first_bad;
    for (int j = threadIdx.x; j < T1p; j += blockDim.x) {
        unsigned long long key = 0ull;
        if (j < T1) {
            // fold the j-parts of k_scores_rowmax in ascending order; strict > keeps the first maximum
            float pv[MAX_WJ];
#pragma unroll
            for (int p = 0; p < MAX_WJ; ++p) pv[p] = (p < nparts) ? pm[(int64_t)p * T1 + j] : -INFINITY;
            float best = pv[0];
#pragma unroll
            for (int p = 1; p < MAX_WJ; ++p) best = pv[p] > best ? pv[p] : best;
            if (badA && (badA[(int64_t)g * T1 + j] ? jmin < T2 : first_bad != 0x7fffffff)) best = __builtin_nanf("");
            if (class_token && j == 0) best = -INFINITY;  // merge.py:59-60: the class token's row is -inf
            key = ((unsigned long long)sort_key(best) << 32) | (0xFFFFFFFFu - (uint32_t)j);
        }
        keys[j] = key;
    }
    __syncthreads();
    const int part = threadIdx.x & 3;
    const unsigned long long ki = keys[i < T1 ? i : T1 - 1];
    int cnt = 0;
    const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(keys + part * quarter);
    for (int j2 = 0; j2 < (quarter >> 1); ++j2) {
        const ulonglong2 k = k2[j2];
        cnt += (k.x > ki) ? 1 : 0;
        cnt += (k.y > ki) ? 1 : 0;
    }
    cnt += __shfl_xor(cnt, 1);
    cnt += __shfl_xor(cnt, 2);
    if (part != 0 || i >= T1) return;
    // this row's own maximum and its first argmax
    float pv[MAX_WJ];
    int pj[MAX_WJ];
#pragma unroll
    for (int p = 0; p < MAX_WJ; ++p) {
        pv[p] = (p < nparts) ? pm[(int64_t)p * T1 + i] : -INFINITY;
        pj[p] = (p < nparts) ? pi[(int64_t)p * T1 + i] : 0;
    }
    float best = pv[0];
    int bidx = pj[0];
#pragma unroll
    for (int p = 1; p < MAX_WJ; ++p) {
        const bool up = pv[p] > best;
        best = up ? pv[p] : best;
        bidx = up ? pj[p] : bidx;
    }
    if (badA) {
        if (badA[(int64_t)g * T1 + i]) {
            if (jmin < T2) {
                best = __builtin_nanf("");
                bidx = jmin;
            }
        } else if (first_bad != 0x7fffffff) {
            best = __builtin_nanf("");
            bidx = first_bad;
        }
    }
    if (class_token && i == 0) {
        best = -INFINITY;
        bidx = 0;
    }
    const int U = T1 - r;
    const int64_t gi = (int64_t)g * T1 + i;
    if (node_max) node_max[gi] = best;
    if (rank_out) rank_out[gi] = cnt;
    if (cnt < r) {
        src_idx[(int64_t)g * r + cnt] = i;
        dst_idx[(int64_t)g * r + cnt] = bidx;
        if (row_map) row_map[gi] = out_row_dst(bidx, U, distill_token);
    } else if (!class_token) {
        unm_idx[(int64_t)g * U + (cnt - r)] = i;
        if (row_map) row_map[gi] = out_row_unm(cnt - r, distill_token);
    }
}

// merge.py:71-73 -- with a class token unm_idx is sorted ascending: stream compaction of the rows
// whose rank is >= r.  One workgroup per group.
__global__ __launch_bounds__(256) void k_compact_unm(const int *__restrict__ rank, int T1, int r,
                                                     int distill_token, int64_t *__restrict__ unm_idx,
                                                     int *__restrict__ row_map) {
    __shared__ int wave_tot[4];
    __shared__ int base_s;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int U = T1 - r;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < T1; i0 += 256) {
        const int i = i0 + tid;
        const bool keep = (i < T1) && (rank[(int64_t)g * T1 + i] >= r);
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (keep) {
            const int pos = off + before;
            unm_idx[(int64_t)g * U + pos] = i;
            if (row_map) row_map[(int64_t)g * T1 + i] = out_row_unm(pos, distill_token);
        }
        __syncthreads();
        if (tid == 0) base_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
}

__global__ void k_edge_keep(const float *__restrict__ node_max, const int64_t *__restrict__ src_idx, int n,
                            int T1, int r, float threshold, uint8_t *__restrict__ keep) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * r) return;
    const int g = (int)(e / r);
    keep[e] = node_max[(int64_t)g * T1 + src_idx[e]] >= threshold ? 1 : 0;
}

