// tome_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ToMe merge path and the
// C ABI declared in include/tome_hip.h.  Built with: hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off -shared -fPIC (csrc/build.py).  No torch, no CUDA, no portability layer.
//
// Launch sequence of one matching (tome_match):
//   k_unit_rows      metric -> fp32 unit vectors, even/odd split, MFMA-fragment order   (HBM bound)
//   k_scores_rowmax  A.B^T tile by tile on v_mfma_f32_32x32x2_f32, running row max/argmax in
//                    registers; the [T1,T2] score matrix never exists in memory           (MFMA bound)
//   k_rank_select    stable descending rank of node_max by counting, writes src/dst/unm   (tiny)
//   k_compact_unm    class-token case only: unm_idx in ascending row order (merge.py:71-73)
// and of one merge (tome_merge_wavg / tome_merge / tome_drop / tome_unmerge):
//   k_merge_rows / k_unmerge_rows   one wave per output (input) token row, 16-byte lanes   (HBM bound)
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

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// element types
// ------------------------------------------------------------------------------------------------
struct bf16_t { uint16_t v; };
struct f16_t { _Float16 v; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return __uint_as_float(((uint32_t)x.v) << 16); }
__device__ __forceinline__ float to_f32(f16_t x) { return (float)x.v; }

template <typename T> __device__ __forceinline__ T from_f32(float f);
template <> __device__ __forceinline__ float from_f32<float>(float f) { return f; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) {
    // round-to-nearest-even, NaN stays NaN (v_cvt_pk_bf16_f32 on gfx950)
    __bf16 b = (__bf16)f;
    bf16_t r;
    __builtin_memcpy(&r.v, &b, 2);
    return r;
}
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float f) {
    f16_t r;
    r.v = (_Float16)f;
    return r;
}

// A lane's slice of a row: VEC consecutive elements moved with one 16-byte (or narrower) access.
template <typename T, int VEC> struct Pack { T e[VEC]; };

template <typename T, int VEC>
__device__ __forceinline__ void load_pack(const T *p, float (&out)[VEC]) {
    typedef Pack<T, VEC> __attribute__((aligned(sizeof(T) * VEC))) P;
    P v = *reinterpret_cast<const P *>(p);
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = to_f32(v.e[i]);
}

template <typename T, int VEC>
__device__ __forceinline__ void store_pack(T *p, const float (&in)[VEC]) {
    typedef Pack<T, VEC> __attribute__((aligned(sizeof(T) * VEC))) P;
    P v;
#pragma unroll
    for (int i = 0; i < VEC; ++i) v.e[i] = from_f32<T>(in[i]);
    *reinterpret_cast<P *>(p) = v;
}

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
template <typename T>
__global__ __launch_bounds__(256) void k_unit_rows_heads(const T *__restrict__ keys, int64_t stride_n,
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
    const int g = (int)(tk / T_);
    const int t = (int)(tk - (int64_t)g * T_);
    const T *row = keys + (int64_t)g * stride_n + (int64_t)t * stride_t + 8 * b8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll 4
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

template <bool ONE_CHUNK>
__global__ __launch_bounds__(64) void k_scores_rowmax(const f32x4 *__restrict__ unitA,
                                                      const f32x4 *__restrict__ unitB, int n, int T1, int T2,
                                                      int nchunk, int ntA, int ntB, int WJ, int64_t groupA_f4,
                                                      int64_t groupB_f4, int distill_token,
                                                      float *__restrict__ part_max, int *__restrict__ part_idx) {
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

    const int lane = threadIdx.x;
    const int col = lane & 31, h = lane >> 5;
    const int i = ti * TILE_ROWS + col;
    // this wave's B tiles: [jt0, jt1), balanced split of ntB over WJ parts
    const int jt0 = (int)(((int64_t)ntB * part) / WJ), jt1 = (int)(((int64_t)ntB * (part + 1)) / WJ);

    const f32x4 *atile = unitA + (int64_t)g * groupA_f4 + frag_index(ti, nchunk, 0, 0, lane);
    const f32x4 *bstream = unitB + (int64_t)g * groupB_f4 + frag_index(jt0, nchunk, 0, 0, lane);

    RowBest rb = {-INFINITY, 0};
    const int nstep = (jt1 - jt0) * nchunk;  // consecutive (tile, chunk) blocks of 512 float4
    if (nstep > 0) {
        f32x4 af[8], bt[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            af[q] = atile[q * 64];
            bt[q] = bstream[q * 64];
        }
        f32x16 acc;
        int jt = jt0, c = 0;
        for (int step = 0; step < nstep; ++step) {
            // operands of the NEXT step replace each float4 right after its last use (single register
            // buffer, each load has 7/8 of a step to land); the last step re-reads its own block
            const int nx = step + 1 < nstep ? step + 1 : step;
            const f32x4 *nb = bstream + (int64_t)nx * 512;
            const int cn = (c + 1 == nchunk) ? 0 : c + 1;
            if (c == 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 b = bt[q], a = af[q];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc, 0, 0, 0);
                bt[q] = nb[q * 64];
                if (!ONE_CHUNK) af[q] = atile[(cn * 8 + q) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c == nchunk - 1) fold_tile(acc, rb, jt, h, T2, distill_token);
            c = cn;
            jt += (cn == 0);
        }
    }
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

__device__ __forceinline__ int out_row_unm(int k, int distill) { return (distill && k >= 1) ? k + 1 : k; }
__device__ __forceinline__ int out_row_dst(int j, int U, int distill) {
    if (!distill) return U + j;
    return j == 0 ? 1 : U + j;
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

// ------------------------------------------------------------------------------------------------
// k_merge_rows: merge.py:75-85 (+ :365-368 when OP_WAVG).  One wave per OUTPUT row; the wave finds
// the sources of a destination row by ballot-scanning dst_idx (r entries, rank order), so the sum is
// atomic-free and in the contract's order.  Every input row is read once, every output row written once.
// ------------------------------------------------------------------------------------------------
enum { OP_WAVG = 100, OP_DROP = 101 };

template <int OP> __device__ __forceinline__ float reduce_step(float acc, float v) {
    if (OP == TOME_SUM || OP == TOME_MEAN || OP == OP_WAVG) return __fadd_rn(acc, v);
    if (OP == TOME_PROD) return __fmul_rn(acc, v);
    if (OP == TOME_AMAX) return (acc != acc) ? acc : (!(v <= acc) ? v : acc);
    if (OP == TOME_AMIN) return (acc != acc) ? acc : (!(v >= acc) ? v : acc);
    return acc;
}

// Where the token rows of group g live.  Contiguous [n,T,C] is {0, T*C, 0, C, 1}; the regrouped views of
// TimeSformer / Motionformer ('b (p t) m -> (b t) p m', timesformer.py:89-90; 'b (s f) d -> (b f) s d',
// motionformer.py:150-151) are {cls*C, (cls+P*F)*C, C, F*C, F}: no permuted copy of x is ever made.
struct TokLayout {
    int64_t base, outer_stride, inner_stride, tok_stride;  // elements
    int inner;                                             // groups per outer index
};

template <typename TX> __device__ __forceinline__ TX *group_ptr(TX *p, const TokLayout &L, int g) {
    return p + L.base + (int64_t)(g / L.inner) * L.outer_stride + (int64_t)(g % L.inner) * L.inner_stride;
}

// One destination row (odd token 2j+1 plus every source merged into it), whole wave, contract order:
// own term first, then the sources in src_idx (rank) order found by ballot-scanning dst_idx.
// Two shapes of the same arithmetic:
//   * rows of at most 2*64 lane-chunks (C <= 1024 bf16 / 512 fp32 with 16-byte lanes): the sources of one
//     64-rank block are compacted onto lanes 0..nsrc-1 (ds_permute), their tokens and sizes fetched as ONE
//     vector load each, and their row chunks fetched four sources at a time before the sequential adds --
//     a destination with k sources costs ~2 + k/4 dependent memory round trips instead of ~2k;
//   * anything wider: the plain sequential loop.
template <typename TX, typename TS, int VEC, int OP>
__device__ __forceinline__ void merge_dst_row(const TX *__restrict__ xg, const TS *__restrict__ sg, int C,
                                              int64_t tstride, int r, int g, int j,
                                              const int64_t *__restrict__ srcg,
                                              const int64_t *__restrict__ dstg, const uint8_t *__restrict__ keep,
                                              TX *__restrict__ orow, TS *__restrict__ srow, int lane) {
    const int t = 2 * j + 1;
    const TX *xr = xg + (int64_t)t * tstride;
    float s_own = 1.0f;
    if (OP == OP_WAVG) s_own = sg ? to_f32(sg[t]) : 1.0f;
    const bool narrow_row = C <= 2 * WAVE * VEC;
    constexpr int NB = VEC >= 8 ? 2 : 4;  // source rows fetched together (register budget)

    if (narrow_row) {
        const int c0 = lane * VEC, c1 = (WAVE + lane) * VEC;
        const bool a0 = c0 < C, a1 = c1 < C;
        float acc0[VEC], acc1[VEC];
        if (a0) load_pack<TX, VEC>(xr + c0, acc0);
        if (a1) load_pack<TX, VEC>(xr + c1, acc1);
        // hybrid: does any incoming edge fall below the threshold (merge.py:326)?
        bool kill = false;
        if (keep && OP != OP_DROP) {
            for (int base = 0; base < r; base += WAVE) {
                const int k = base + lane;
                const bool m = (k < r) && ((int)dstg[k] == j);
                kill = kill || (__ballot(m && keep[(int64_t)g * r + k] == 0) != 0ull);
            }
        }
        float ssum = s_own;
        int cnt = 1;
        if (OP == OP_WAVG) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (a0) acc0[e] = __fmul_rn(acc0[e], s_own);
                if (a1) acc1[e] = __fmul_rn(acc1[e], s_own);
            }
        }
        if (kill) {
            if (OP == OP_WAVG) ssum = __fmul_rn(ssum, 0.0f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (a0) acc0[e] = __fmul_rn(acc0[e], 0.0f);
                if (a1) acc1[e] = __fmul_rn(acc1[e], 0.0f);
            }
        }
        if (OP != OP_DROP) {
            for (int base = 0; base < r; base += WAVE) {
                const int k = base + lane;
                const bool m = (k < r) && ((int)dstg[k] == j);
                const unsigned long long mk = __ballot(m);
                if (mk == 0ull) continue;
                const int nsrc = __popcll(mk);
                // lane L < nsrc learns the lane (= rank offset) of the L-th source of this block
                const int myrank = __popcll(mk & ((1ull << lane) - 1ull));
                const int kL = __builtin_amdgcn_ds_permute(m ? myrank * 4 : 63 * 4 + 256, m ? lane : 0);
                int tsL = 0;
                float sL = 1.0f;
                if (lane < nsrc) {
                    tsL = 2 * (int)srcg[base + kL];
                    if (OP == OP_WAVG && sg) sL = to_f32(sg[tsL]);
                }
                for (int e0 = 0; e0 < nsrc; e0 += NB) {
                    typedef Pack<TX, VEC> __attribute__((aligned(sizeof(TX) * VEC))) PK;
                    PK p0[NB], p1[NB];  // raw chunks of up to NB source rows, all in flight together
                    float sq[NB];
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const int e = (e0 + u < nsrc) ? e0 + u : nsrc - 1;
                        const int tsu = __builtin_amdgcn_readlane(tsL, e);
                        sq[u] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sL), e));
                        const TX *sr = xg + (int64_t)tsu * tstride;
                        if (a0) p0[u] = *reinterpret_cast<const PK *>(sr + c0);
                        if (a1) p1[u] = *reinterpret_cast<const PK *>(sr + c1);
                    }
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        if (e0 + u < nsrc) {
#pragma unroll
                            for (int e = 0; e < VEC; ++e) {
                                if (a0) {
                                    const float v = to_f32(p0[u].e[e]);
                                    acc0[e] = reduce_step<OP>(acc0[e], (OP == OP_WAVG) ? __fmul_rn(v, sq[u]) : v);
                                }
                                if (a1) {
                                    const float v = to_f32(p1[u].e[e]);
                                    acc1[e] = reduce_step<OP>(acc1[e], (OP == OP_WAVG) ? __fmul_rn(v, sq[u]) : v);
                                }
                            }
                            if (OP == OP_WAVG) ssum = __fadd_rn(ssum, sq[u]);
                            ++cnt;
                        }
                    }
                }
            }
        }
        if (OP == OP_WAVG) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (a0) acc0[e] = __fdiv_rn(acc0[e], ssum);
                if (a1) acc1[e] = __fdiv_rn(acc1[e], ssum);
            }
        } else if (OP == TOME_MEAN && cnt > 1) {
            const float fc = (float)cnt;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (a0) acc0[e] = __fdiv_rn(acc0[e], fc);
                if (a1) acc1[e] = __fdiv_rn(acc1[e], fc);
            }
        }
        if (a0) store_pack<TX, VEC>(orow + c0, acc0);
        if (a1) store_pack<TX, VEC>(orow + c1, acc1);
        if (OP == OP_WAVG && lane == 0) *srow = from_f32<TS>(ssum);
        return;
    }

    // wide rows: sequential form
    unsigned long long mask0 = 0ull;
    bool kill = false;
    if (OP != OP_DROP) {
        for (int base = 0; base < r; base += WAVE) {
            const int k = base + lane;
            const bool m = (k < r) && ((int)dstg[k] == j);
            const unsigned long long mk = __ballot(m);
            if (base == 0) mask0 = mk;
            if (keep) kill = kill || (__ballot(m && keep[(int64_t)g * r + k] == 0) != 0ull);
        }
    }
    float ssum = s_own;
    int cnt = 1;
    if (OP == OP_WAVG) {
        if (kill) ssum = __fmul_rn(ssum, 0.0f);
    }
    bool first_chunk = true;
    for (int c0 = 0; c0 < C; c0 += WAVE * VEC) {
        const int c = c0 + lane * VEC;
        const bool act = c < C;
        float acc[VEC];
        if (act) {
            load_pack<TX, VEC>(xr + c, acc);
            if (OP == OP_WAVG) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = __fmul_rn(acc[e], s_own);
            }
            if (kill) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = __fmul_rn(acc[e], 0.0f);
            }
        }
        if (OP != OP_DROP) {
            for (int base = 0; base < r; base += WAVE) {
                unsigned long long mk;
                if (base == 0) mk = mask0;
                else {
                    const int k = base + lane;
                    mk = __ballot((k < r) && ((int)dstg[k] == j));
                }
                while (mk) {
                    const int b = __ffsll((long long)mk) - 1;
                    mk &= mk - 1ull;
                    const int ts = 2 * (int)srcg[base + b];
                    float s2 = 1.0f;
                    if (OP == OP_WAVG) s2 = sg ? to_f32(sg[ts]) : 1.0f;
                    if (act) {
                        float v[VEC];
                        load_pack<TX, VEC>(xg + (int64_t)ts * tstride + c, v);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            float p = (OP == OP_WAVG) ? __fmul_rn(v[e], s2) : v[e];
                            acc[e] = reduce_step<OP>(acc[e], p);
                        }
                    }
                    if (first_chunk) {
                        if (OP == OP_WAVG) ssum = __fadd_rn(ssum, s2);
                        ++cnt;
                    }
                }
            }
        }
        first_chunk = false;
        if (act) {
            if (OP == OP_WAVG) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = __fdiv_rn(acc[e], ssum);
            } else if (OP == TOME_MEAN) {
                if (cnt > 1) {
                    const float fc = (float)cnt;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = __fdiv_rn(acc[e], fc);
                }
            }
            store_pack<TX, VEC>(orow + c, acc);
        }
    }
    if (OP == OP_WAVG && lane == 0) *srow = from_f32<TS>(ssum);
}

// inverse of the output layout (merge.py:82-85): output row o -> (is it a B/dst row?, index in its set)
__device__ __forceinline__ void decode_out_row(int o, int U, int distill, bool &is_dst, int &idx) {
    if (!distill) {
        is_dst = o >= U;
        idx = is_dst ? o - U : o;
    } else if (o == 0) { is_dst = false; idx = 0; }
    else if (o == 1) { is_dst = true; idx = 0; }
    else if (o <= U) { is_dst = false; idx = o - 1; }
    else { is_dst = true; idx = o - U; }
}

// Generic form: one wave per output row, any C / alignment / r.  (The hot shapes go through
// k_merge_rows_fast below; this one serves odd channel counts such as the size column or source matrices.)
template <typename TX, typename TS, int VEC, int OP>
__global__ __launch_bounds__(256) void k_merge_rows(const TX *__restrict__ x, const TS *__restrict__ size,
                                                    int n, int T_, int C, int r,
                                                    const int64_t *__restrict__ src_idx,
                                                    const int64_t *__restrict__ dst_idx,
                                                    const int64_t *__restrict__ unm_idx, int distill,
                                                    const uint8_t *__restrict__ keep, TX *__restrict__ xout,
                                                    TS *__restrict__ sout, TokLayout lin, TokLayout lout) {
    const int lane = threadIdx.x & 63;
    const int To = T_ - r;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= (int64_t)n * To) return;
    const int g = (int)(row / To);
    const int o = (int)(row - (int64_t)g * To);
    const int T1 = (T_ + 1) >> 1, U = T1 - r;
    bool is_dst;
    int idx;
    decode_out_row(o, U, distill, is_dst, idx);

    const TX *xg = group_ptr(x, lin, g);
    const TS *sg = size ? size + (int64_t)g * T_ : nullptr;
    TX *orow = group_ptr(xout, lout, g) + (int64_t)o * lout.tok_stride;
    if (!is_dst) {
        const int t = 2 * (int)unm_idx[(int64_t)g * U + idx];
        const TX *xr = xg + (int64_t)t * lin.tok_stride;
        float s = 1.0f;
        if (OP == OP_WAVG) s = sg ? to_f32(sg[t]) : 1.0f;
        for (int c = lane * VEC; c < C; c += WAVE * VEC) {
            float v[VEC];
            load_pack<TX, VEC>(xr + c, v);
            if (OP == OP_WAVG) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = __fdiv_rn(__fmul_rn(v[e], s), s);
            }
            store_pack<TX, VEC>(orow + c, v);
        }
        if (OP == OP_WAVG && lane == 0) sout[row] = from_f32<TS>(s);
        return;
    }
    merge_dst_row<TX, TS, VEC, OP>(xg, sg, C, lin.tok_stride, r, g, idx, src_idx ? src_idx + (int64_t)g * r : nullptr,
                                   dst_idx ? dst_idx + (int64_t)g * r : nullptr, keep, orow,
                                   sout ? sout + row : nullptr, lane);
}

// ------------------------------------------------------------------------------------------------
// k_merge_rows_fast: the HBM-bound form for rows made of whole 16-byte chunks (C*sizeof(TX) % 16 == 0,
// at most 384 chunks per R rows).  A wave owns R consecutive OUTPUT rows of one group, flattens their
// R*cpr 16-byte chunks over its lanes (6 chunks per lane, every lane busy, stores contiguous across the
// R rows) and issues all of its loads before touching any of them, so ~6 KiB per wave are in flight.
// Rows that nothing merges into are moved as raw bits when their size is 1 ((x*1)/1 == x bit for bit)
// or scaled in fp32 otherwise; the few rows that receive sources are finished by merge_dst_row.
// ------------------------------------------------------------------------------------------------
#define FAST_NIT 6
#define FAST_MAXR 4

template <typename TX, typename TS, int OP, int NIT>
__global__ __launch_bounds__(256) void k_merge_rows_fast(const TX *__restrict__ x, const TS *__restrict__ size,
                                                         int n, int T_, int C, int r, int R, int cpr,
                                                         const int64_t *__restrict__ src_idx,
                                                         const int64_t *__restrict__ dst_idx,
                                                         const int64_t *__restrict__ unm_idx, int distill,
                                                         const uint8_t *__restrict__ keep, TX *__restrict__ xout,
                                                         TS *__restrict__ sout, TokLayout lin, TokLayout lout,
                                                         int cls_rows) {
    constexpr int VEC = 16 / sizeof(TX);
    const int lane = threadIdx.x & 63;
    const int To = T_ - r;
    const int rg_per_group = (To + R - 1) / R;
    const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_main = (int64_t)n * rg_per_group;
    const int64_t n_edge = (OP == OP_DROP) ? 0 : (int64_t)n * r;
    if (w >= n_main + n_edge) {
        // the class tokens kept aside by the regrouped callers (timesformer.py:89,107): plain row copies
        const int64_t b = w - n_main - n_edge;
        if (b < cls_rows) {
            const uint4 *src = reinterpret_cast<const uint4 *>(x + b * lin.outer_stride);
            uint4 *dst = reinterpret_cast<uint4 *>(xout + b * lout.outer_stride);
            for (int c = lane; c < cpr; c += WAVE) dst[c] = src[c];
        }
        return;
    }
    if (w >= n_main) {
        // edge waves: one per (group, rank k).  The wave of the FIRST edge into a destination builds that
        // row (all its sources, rank order); the others leave.  Destinations with sources therefore never
        // hold up the streaming waves above, and run concurrently with them.
        const int64_t ew = w - n_main;
        const int g = (int)(ew / r);
        const int k = (int)(ew - (int64_t)g * r);
        const int64_t *dstg = dst_idx + (int64_t)g * r;
        const int j = (int)dstg[k];
        bool earlier = false;
        for (int base = 0; base < k; base += WAVE) {
            const int kk = base + lane;
            earlier = earlier || (__ballot((kk < k) && ((int)dstg[kk] == j)) != 0ull);
        }
        if (earlier) return;
        const int T1e = (T_ + 1) >> 1, Ue = T1e - r;
        const int o = out_row_dst(j, Ue, distill);
        merge_dst_row<TX, TS, VEC, OP>(group_ptr(x, lin, g), size ? size + (int64_t)g * T_ : nullptr, C, lin.tok_stride,
                                       r, g, j, src_idx + (int64_t)g * r, dstg, keep,
                                       group_ptr(xout, lout, g) + (int64_t)o * lout.tok_stride,
                                       sout ? sout + (int64_t)g * To + o : nullptr, lane);
        return;
    }
    const int g = (int)(w / rg_per_group);
    const int o0 = (int)(w - (int64_t)g * rg_per_group) * R;
    const int T1 = (T_ + 1) >> 1, U = T1 - r;
    const TX *xg = group_ptr(x, lin, g);
    TX *og = group_ptr(xout, lout, g);
    const TS *sg = size ? size + (int64_t)g * T_ : nullptr;
    const int64_t *dstg = dst_idx ? dst_idx + (int64_t)g * r : nullptr;

    // per-row facts, computed by lanes 0..R-1 in parallel and broadcast as wave-uniform scalars: source
    // token, B-row number, "something merges into it", size.  Load order matters for latency: the first block
    // of dst_idx and the unm_idx entries go out together; the sizes are requested as soon as the tokens are
    // known but only READ after the token rows' own loads have been issued, so a wave waits for two memory
    // round trips (index -> rows), not three.
    const int d_first = (OP != OP_DROP && lane < r) ? (int)dstg[lane] : -2;
    int my_tok = 0, my_j = -1;
    float my_s = 1.0f;
    bool my_valid = false;
    if (lane < R) {
        const int o = o0 + lane;
        if (o < To) {
            my_valid = true;
            bool is_dst;
            int idx;
            decode_out_row(o, U, distill, is_dst, idx);
            if (is_dst) {
                my_tok = 2 * idx + 1;
                my_j = idx;
            } else {
                my_tok = 2 * (int)unm_idx[(int64_t)g * U + idx];
            }
        }
    }
    TS my_s_raw;
    const bool load_size = (OP == OP_WAVG) && sg && my_valid;
    if (load_size) my_s_raw = sg[my_tok];
    const unsigned long long vmask = __ballot(my_valid);
    const int tok0 = __builtin_amdgcn_readlane(my_tok, 0), tok1 = __builtin_amdgcn_readlane(my_tok, 1),
              tok2 = __builtin_amdgcn_readlane(my_tok, 2), tok3 = __builtin_amdgcn_readlane(my_tok, 3);
    const int j0 = __builtin_amdgcn_readlane(my_j, 0), j1 = __builtin_amdgcn_readlane(my_j, 1),
              j2 = __builtin_amdgcn_readlane(my_j, 2), j3 = __builtin_amdgcn_readlane(my_j, 3);
    bool e0 = false, e1 = false, e2 = false, e3 = false;  // rows that receive sources
    if (OP != OP_DROP) {
        e0 = __ballot(d_first == j0) != 0ull;
        e1 = __ballot(d_first == j1) != 0ull;
        e2 = __ballot(d_first == j2) != 0ull;
        e3 = __ballot(d_first == j3) != 0ull;
        for (int base = WAVE; base < r; base += WAVE) {
            const int k = base + lane;
            const int d = (k < r) ? (int)dstg[k] : -2;
            e0 = e0 || (__ballot(d == j0) != 0ull);
            e1 = e1 || (__ballot(d == j1) != 0ull);
            e2 = e2 || (__ballot(d == j2) != 0ull);
            e3 = e3 || (__ballot(d == j3) != 0ull);
        }
    }
    const bool ok0 = (vmask & 1ull) && !e0, ok1 = (vmask & 2ull) && !e1, ok2 = (vmask & 4ull) && !e2,
               ok3 = (vmask & 8ull) && !e3;

    // flattened chunk loop: chunk q of the R-row slab -> (row q / cpr, 16-byte column q % cpr)
    const int total = R * cpr;
    uint4 raw[NIT];
    int rowof[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = it * WAVE + lane;
        const int rr = (q >= cpr) + (q >= 2 * cpr) + (q >= 3 * cpr);
        const int cc = q - rr * cpr;
        const int t = rr == 0 ? tok0 : (rr == 1 ? tok1 : (rr == 2 ? tok2 : tok3));
        const bool ok = (rr == 0 ? ok0 : (rr == 1 ? ok1 : (rr == 2 ? ok2 : ok3))) && (q < total);
        rowof[it] = ok ? rr : -1;
        if (ok)
            raw[it] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(xg + (int64_t)t * lin.tok_stride) +
                                                       cc * 16);
    }
    if (load_size) my_s = to_f32(my_s_raw);
    const float sz0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_s), 0)),
                sz1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_s), 1)),
                sz2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_s), 2)),
                sz3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_s), 3));
    // (x*s)/s is x itself when s is 1, and also when s is a power of two and x came from a 16-bit
    // format (the fp32 product cannot overflow): those rows move as raw bits
    constexpr bool narrow = sizeof(TX) == 2;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rr = rowof[it];
        if (rr < 0) continue;
        const int q = it * WAVE + lane;
        const float s = rr == 0 ? sz0 : (rr == 1 ? sz1 : (rr == 2 ? sz2 : sz3));
        uint4 outv = raw[it];
        if (OP == OP_WAVG) {
            const uint32_t sb = __float_as_uint(s);
            const bool exact = (s == 1.0f) || (narrow && (sb & 0x007FFFFFu) == 0u && s >= 1.0f && s <= 65536.0f);
            if (!exact) {
                Pack<TX, VEC> pk;
                __builtin_memcpy(&pk, &raw[it], 16);
#pragma unroll
                for (int e = 0; e < VEC; ++e) pk.e[e] = from_f32<TX>(__fdiv_rn(__fmul_rn(to_f32(pk.e[e]), s), s));
                __builtin_memcpy(&outv, &pk, 16);
            }
        }
        const int cc = q - rr * cpr;
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(og + (int64_t)(o0 + rr) * lout.tok_stride) + cc * 16) = outv;
    }
    if (OP == OP_WAVG && lane < R && my_valid) {
        const bool mine_has_edges = lane == 0 ? e0 : (lane == 1 ? e1 : (lane == 2 ? e2 : e3));
        if (!mine_has_edges) sout[(int64_t)g * To + o0 + lane] = from_f32<TS>(my_s);
    }
}

// k_unmerge_rows: merge.py:87-100 as a scatter from the merged sequence: one wave per INPUT row; a
// destination row also lands on every even slot that was merged into it.  src and unm partition the
// even slots, so every output row is written exactly once and no zero fill is needed.
template <typename TX, int VEC>
__global__ __launch_bounds__(256) void k_unmerge_rows(const TX *__restrict__ x, int n, int T_, int C, int r,
                                                      const int64_t *__restrict__ src_idx,
                                                      const int64_t *__restrict__ dst_idx,
                                                      const int64_t *__restrict__ unm_idx,
                                                      TX *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int To = T_ - r;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= (int64_t)n * To) return;
    const int g = (int)(row / To);
    const int o = (int)(row - (int64_t)g * To);
    const int T1 = (T_ + 1) >> 1, U = T1 - r;
    const TX *xr = x + row * C;
    TX *og = out + (int64_t)g * T_ * C;
    if (o < U) {
        TX *dst = og + (int64_t)(2 * (int)unm_idx[(int64_t)g * U + o]) * C;
        for (int c = lane * VEC; c < C; c += WAVE * VEC) {
            float v[VEC];
            load_pack<TX, VEC>(xr + c, v);
            store_pack<TX, VEC>(dst + c, v);
        }
        return;
    }
    const int j = o - U;
    const int64_t *srcg = src_idx + (int64_t)g * r, *dstg = dst_idx + (int64_t)g * r;
    for (int c = lane * VEC; c < C; c += WAVE * VEC) {
        float v[VEC];
        load_pack<TX, VEC>(xr + c, v);
        store_pack<TX, VEC>(og + (int64_t)(2 * j + 1) * C + c, v);
    }
    for (int base = 0; base < r; base += WAVE) {
        const int k = base + lane;
        unsigned long long mk = __ballot((k < r) && ((int)dstg[k] == j));
        while (mk) {
            const int b = __ffsll((long long)mk) - 1;
            mk &= mk - 1ull;
            TX *dst = og + (int64_t)(2 * (int)srcg[base + b]) * C;
            for (int c = lane * VEC; c < C; c += WAVE * VEC) {
                float v[VEC];
                load_pack<TX, VEC>(xr + c, v);
                store_pack<TX, VEC>(dst + c, v);
            }
        }
    }
}

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

// Optional per-stage timing of tome_match (bench.py's roofline figures): events are created when
// profiling is switched on, never inside a launch path.
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

struct MatchWs {
    float *unitA, *unitB, *part_max;
    int *part_idx, *rank;
    uint8_t *badA, *badB;
    int ntA, ntB, nchunk;
    int64_t groupA_f4, groupB_f4;  // float4 per group of each unit set
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

// shared tail of tome_match / tome_match_keys: stages 2 (similarity + row max) and 3 (rank + select)
static int match_tail(const MatchWs &w, int64_t n, int64_t T, int64_t re, int class_token, int distill_token,
                      int64_t *src_idx, int64_t *dst_idx, int64_t *unm_idx, float *node_max, int32_t *row_map,
                      hipStream_t st) {
    const int T1 = (int)((T + 1) / 2), T2 = (int)(T / 2);
    const int prof_reps = g_prof.on ? g_prof.reps : 1;
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
    for (int rep = 0; rep < prof_reps; ++rep)
    if (w.nchunk == 1)
        hipLaunchKernelGGL(k_scores_rowmax<true>, dim3(nb2), dim3(64), 0, st, (const f32x4 *)w.unitA,
                           (const f32x4 *)w.unitB, (int)n, T1, T2, w.nchunk, w.ntA, w.ntB, WJ, w.groupA_f4,
                           w.groupB_f4, distill_token, w.part_max, w.part_idx);
    else
        hipLaunchKernelGGL(k_scores_rowmax<false>, dim3(nb2), dim3(64), 0, st, (const f32x4 *)w.unitA,
                           (const f32x4 *)w.unitB, (int)n, T1, T2, w.nchunk, w.ntA, w.ntB, WJ, w.groupA_f4,
                           w.groupB_f4, distill_token, w.part_max, w.part_idx);
    if (int rc = check_launch("k_scores_rowmax")) return rc;
    prof_mark(2, st);

    // 3. rank + select
    int rc = TOME_OK;
    for (int rep = 0; rep < prof_reps && rc == TOME_OK; ++rep)
        rc = launch_select(w, WJ, true, n, T, re, class_token, distill_token, src_idx, dst_idx, unm_idx, node_max, row_map, st);
    prof_mark(3, st);
    g_prof.valid = g_prof.on && rc == TOME_OK;
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
    bool launched = false;
    const int prof_reps = g_prof.on ? g_prof.reps : 1;
    for (int rep = 0; rep < prof_reps; ++rep) {
    launched = false;
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

    return match_tail(w, n, T, re, class_token, distill_token, src_idx, dst_idx, unm_idx, node_max, row_map, st);
}

extern "C" int tome_match_keys(const void *keys, int dtype, int64_t n, int64_t H, int64_t T, int64_t D,
                               int64_t stride_n, int64_t stride_h, int64_t stride_t, int64_t r, int class_token,
                               int distill_token, int64_t *src_idx, int64_t *dst_idx, int64_t *unm_idx,
                               float *node_max, int32_t *row_map, void *workspace, size_t workspace_bytes,
                               tome_stream_t stream) {
    if (!keys || n <= 0 || T <= 0 || H <= 0) return fail(TOME_EINVAL, "tome_match_keys: bad shape/pointer");
    if (D != 64) return fail(TOME_EINVAL, "tome_match_keys: head dimension %lld (only 64 is fused)", (long long)D);
    if (n > 0x7fffffff / T) return fail(TOME_EINVAL, "tome_match_keys: problem too large");
    const size_t es = dtype == TOME_F32 ? 4 : 2;
    if (dtype < TOME_F32 || dtype > TOME_F16) return fail(TOME_EINVAL, "tome_match_keys: dtype %d", dtype);
    if (((uintptr_t)keys) % 16 || (stride_n * es) % 16 || (stride_h * es) % 16 || (stride_t * es) % 16)
        return fail(TOME_EINVAL, "tome_match_keys: keys must be 16-byte aligned in every stride");
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
    const int prof_reps = g_prof.on ? g_prof.reps : 1;
    const unsigned nb = (unsigned)((n * T + 31) / 32);
    for (int rep = 0; rep < prof_reps; ++rep) {
        switch (dtype) {
        case TOME_F32:
            hipLaunchKernelGGL(k_unit_rows_heads<float>, dim3(nb), dim3(256), 0, st, (const float *)keys, stride_n,
                               stride_h, stride_t, (int)n, (int)H, (int)T, w.unitA, w.unitB, w.groupA_f4, w.groupB_f4,
                               w.badA, w.badB);
            break;
        case TOME_BF16:
            hipLaunchKernelGGL(k_unit_rows_heads<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t *)keys, stride_n,
                               stride_h, stride_t, (int)n, (int)H, (int)T, w.unitA, w.unitB, w.groupA_f4, w.groupB_f4,
                               w.badA, w.badB);
            break;
        default:
            hipLaunchKernelGGL(k_unit_rows_heads<f16_t>, dim3(nb), dim3(256), 0, st, (const f16_t *)keys, stride_n,
                               stride_h, stride_t, (int)n, (int)H, (int)T, w.unitA, w.unitB, w.groupA_f4, w.groupB_f4,
                               w.badA, w.badB);
            break;
        }
    }
    if (int rc = check_launch("k_unit_rows_heads")) return rc;
    prof_mark(1, st);
    return match_tail(w, n, T, re, class_token, distill_token, src_idx, dst_idx, unm_idx, node_max, row_map, st);
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
                             const TokLayout *lin_p = nullptr, const TokLayout *lout_p = nullptr, int cls_rows = 0) {
    constexpr int VEC = 16 / sizeof(TX);
    const int64_t To = T - r;
    const TokLayout lin = lin_p ? *lin_p : contiguous_layout(T, C);
    const TokLayout lout = lout_p ? *lout_p : contiguous_layout(To, C);
    const bool vec_ok = (C % VEC == 0) && aligned16(x) && aligned16(xout);
    const int64_t cpr = C / VEC;  // 16-byte chunks per row
    if (vec_ok && cpr <= FAST_NIT * WAVE) {
        // rows per wave: measured on MI355X, a plain 16-byte copy runs fastest with ONE load per lane in
        // flight and many waves (6.0-6.9 TB/s) and loses ~10 % at four; NIT=3 (two 1536-byte rows per wave)
        // is the smallest slab that still keeps every lane busy for 768-channel bf16 tokens
        static const int nit_pref = [] {
            const char *e = getenv("TOME_MERGE_NIT");
            int v = e ? atoi(e) : 0;
            return (v == 3 || v == 6) ? v : 6;
        }();
        const int nit = (cpr <= 3 * WAVE) ? nit_pref : FAST_NIT;
        int R = (int)((nit * WAVE) / cpr);
        if (R > FAST_MAXR) R = FAST_MAXR;
        const int64_t waves = n * ((To + R - 1) / R) + (OP == OP_DROP ? 0 : n * r) + cls_rows;
        if (nit == 3)
            hipLaunchKernelGGL((k_merge_rows_fast<TX, TS, OP, 3>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st,
                               (const TX *)x, (const TS *)size, (int)n, (int)T, (int)C, (int)r, R, (int)cpr, src, dst,
                               unm, distill, keep, (TX *)xout, (TS *)sout, lin, lout, cls_rows);
        else
            hipLaunchKernelGGL((k_merge_rows_fast<TX, TS, OP, 6>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st,
                               (const TX *)x, (const TS *)size, (int)n, (int)T, (int)C, (int)r, R, (int)cpr, src, dst,
                               unm, distill, keep, (TX *)xout, (TS *)sout, lin, lout, cls_rows);
        return check_launch("k_merge_rows_fast");
    }
    if (cls_rows) return fail(TOME_EINVAL, "regrouped merge needs rows of whole 16-byte chunks (C=%lld)", (long long)C);
    const int64_t rows = n * To;
    const unsigned nb = (unsigned)((rows + 3) / 4);
    if (vec_ok)
        hipLaunchKernelGGL((k_merge_rows<TX, TS, VEC, OP>), dim3(nb), dim3(256), 0, st, (const TX *)x,
                           (const TS *)size, (int)n, (int)T, (int)C, (int)r, src, dst, unm, distill, keep, (TX *)xout,
                           (TS *)sout, lin, lout);
    else
        hipLaunchKernelGGL((k_merge_rows<TX, TS, 1, OP>), dim3(nb), dim3(256), 0, st, (const TX *)x,
                           (const TS *)size, (int)n, (int)T, (int)C, (int)r, src, dst, unm, distill, keep, (TX *)xout,
                           (TS *)sout, lin, lout);
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
                               void *size_out, tome_stream_t stream) {
    if (int rc = check_merge_args("tome_merge_wavg", x, n, T, C, r, x_out)) return rc;
    if (!src_idx || !dst_idx || (!unm_idx && (T + 1) / 2 > r) || !size_out)
        return fail(TOME_EINVAL, "tome_merge_wavg: null buffer");
    hipStream_t st = (hipStream_t)stream;
#define WAVG(TX, TS)                                                                                         \
    return launch_merge_rows<TX, TS, OP_WAVG>(x, size, n, T, C, r, src_idx, dst_idx, unm_idx, distill_token, \
                                              edge_keep, x_out, size_out, st)
    if (x_dtype == TOME_F32 && size_dtype == TOME_F32) WAVG(float, float);
    if (x_dtype == TOME_BF16 && size_dtype == TOME_BF16) WAVG(bf16_t, bf16_t);
    if (x_dtype == TOME_BF16 && size_dtype == TOME_F32) WAVG(bf16_t, float);
    if (x_dtype == TOME_F16 && size_dtype == TOME_F16) WAVG(f16_t, f16_t);
    if (x_dtype == TOME_F16 && size_dtype == TOME_F32) WAVG(f16_t, float);
#undef WAVG
    return fail(TOME_EINVAL, "tome_merge_wavg: unsupported dtypes x=%d size=%d", x_dtype, size_dtype);
}

extern "C" int tome_merge_wavg_regrouped(const void *x, int x_dtype, const void *size, int size_dtype, int64_t B,
                                         int64_t F, int64_t P, int64_t C, int64_t r, int has_cls,
                                         const int64_t *src_idx, const int64_t *dst_idx, const int64_t *unm_idx,
                                         const uint8_t *edge_keep, void *x_out, void *size_out,
                                         tome_stream_t stream) {
    if (B <= 0 || F <= 0) return fail(TOME_EINVAL, "tome_merge_wavg_regrouped: bad shape");
    const int64_t n = B * F;
    if (int rc = check_merge_args("tome_merge_wavg_regrouped", x, n, P, C, r, x_out)) return rc;
    if (!src_idx || !dst_idx || (!unm_idx && (P + 1) / 2 > r) || !size_out)
        return fail(TOME_EINVAL, "tome_merge_wavg_regrouped: null buffer");
    const int cls = has_cls ? 1 : 0;
    const TokLayout lin{cls * C, (cls + P * F) * C, C, F * C, (int)F};
    const TokLayout lout{cls * C, (cls + (P - r) * F) * C, C, F * C, (int)F};
    hipStream_t st = (hipStream_t)stream;
#define WAVGR(TX, TS)                                                                                         \
    return launch_merge_rows<TX, TS, OP_WAVG>(x, size, n, P, C, r, src_idx, dst_idx, unm_idx, 0, edge_keep, x_out, \
                                              size_out, st, &lin, &lout, cls ? (int)B : 0)
    if (x_dtype == TOME_F32 && size_dtype == TOME_F32) WAVGR(float, float);
    if (x_dtype == TOME_BF16 && size_dtype == TOME_BF16) WAVGR(bf16_t, bf16_t);
    if (x_dtype == TOME_BF16 && size_dtype == TOME_F32) WAVGR(bf16_t, float);
    if (x_dtype == TOME_F16 && size_dtype == TOME_F16) WAVGR(f16_t, f16_t);
    if (x_dtype == TOME_F16 && size_dtype == TOME_F32) WAVGR(f16_t, float);
#undef WAVGR
    return fail(TOME_EINVAL, "tome_merge_wavg_regrouped: unsupported dtypes x=%d size=%d", x_dtype, size_dtype);
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
