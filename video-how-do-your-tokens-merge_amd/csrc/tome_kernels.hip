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
// k_unit_rows: merge.py:51-52.  One thread per token runs the sequential fma chain of the squared
// norm (the contract's order); the division and the scatter into the two fragment-ordered sets
// are spread over the whole workgroup.
//   unitA [n][T1][Dp], unitB [n][T2][Dp]; a row holds its even-k elements in the first Dp/2 floats
//   and its odd-k elements in the second Dp/2 -- lane half h of v_mfma_f32_32x32x2_f32 supplies
//   k = 2s+h at step s, so each lane reads one contiguous run.  Dp = D rounded up to 64, zero filled.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_unit_rows(const T *__restrict__ metric, int64_t stride_n,
                                                   int64_t stride_t, int n, int T_, int D, int Dp,
                                                   int tok_per_wg, float *__restrict__ unitA,
                                                   float *__restrict__ unitB) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ld = D + 1;
    float *tile = lds;                      // [tok_per_wg][D+1]
    float *nrm = lds + (size_t)tok_per_wg * ld;  // [tok_per_wg]
    const int tid = threadIdx.x;
    const int64_t tok0 = (int64_t)blockIdx.x * tok_per_wg;
    const int64_t ntok = (int64_t)n * T_;
    const int T1 = (T_ + 1) >> 1, T2 = T_ >> 1;

    for (int lt = tid >> 6; lt < tok_per_wg; lt += 4) {
        int64_t tok = tok0 + lt;
        if (tok >= ntok) break;
        int64_t g = tok / T_;
        int t = (int)(tok - g * T_);
        const T *row = metric + g * stride_n + (int64_t)t * stride_t;
        for (int k = tid & 63; k < D; k += 64) tile[lt * ld + k] = to_f32(row[k]);
    }
    __syncthreads();
    if (tid < tok_per_wg && tok0 + tid < ntok) {
        const float *v = tile + tid * ld;
        float ss = 0.0f;
        for (int k = 0; k < D; ++k) ss = __fmaf_rn(v[k], v[k], ss);
        nrm[tid] = __fsqrt_rn(ss);
    }
    __syncthreads();
    const int half = Dp >> 1;
    for (int lt = tid >> 6; lt < tok_per_wg; lt += 4) {
        int64_t tok = tok0 + lt;
        if (tok >= ntok) break;
        int64_t g = tok / T_;
        int t = (int)(tok - g * T_);
        float *dst = (t & 1) ? unitB + ((int64_t)g * T2 + (t >> 1)) * Dp
                             : unitA + ((int64_t)g * T1 + (t >> 1)) * Dp;
        float nr = nrm[lt];
        for (int k = tid & 63; k < Dp; k += 64) {
            float u = (k < D) ? __fdiv_rn(tile[lt * ld + k], nr) : 0.0f;
            dst[(k & 1) * half + (k >> 1)] = u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_scores_rowmax: merge.py:53,59-64 without the score matrix.
//   workgroup = W waves (W in {1,2,4,8}); wave w owns 32 A rows (columns of the transposed tile),
//   every wave of the group streams the same B rows through LDS (double buffered, one barrier per
//   32-row tile).  S^T tile = Bhat_tile (MFMA "A" operand, from LDS) x Ahat^T (MFMA "B" operand,
//   registers): accumulator register v of lane l holds S[i = l&31][j = (v&3)+8(v>>2)+4(l>>5)], so
//   the max over j is a per-lane running max, combined across the two lane halves once at the end.
//   v_mfma_f32_32x32x2_f32 adds k = 2s then k = 2s+1 to the accumulator: a k-ordered fma chain.
// ------------------------------------------------------------------------------------------------
#define TILE_ROWS 32
#define LDS_ROW 68 /* floats: 64 + 4 pad -> 272-byte rows, conflict-free ds_read_b128 */

template <bool ONE_CHUNK>
__global__ __launch_bounds__(512) void k_scores_rowmax(const float *__restrict__ unitA,
                                                       const float *__restrict__ unitB, int n, int T1,
                                                       int T2, int Dp, int wg_per_group, int class_token,
                                                       int distill_token, float *__restrict__ node_max,
                                                       int *__restrict__ node_idx) {
    __shared__ __attribute__((aligned(16))) float lds[2][TILE_ROWS * LDS_ROW];
    // XCD-aware block -> (group, tile) map: blocks b and b+8 share an XCD (round-robin dispatch), so
    // all workgroups of one group -- which stream the same B rows -- are given ids congruent mod 8
    // and find those rows in their XCD's L2.  Placement only affects speed.
    const int L = blockIdx.x;
    const int xcd = L & 7, q = L >> 3;
    const int g = (q / wg_per_group) * 8 + xcd;
    const int wg_in_group = q % wg_per_group;
    if (g >= n) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = blockDim.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int i0 = (wg_in_group * W + wave) * TILE_ROWS;
    const int i = i0 + col;
    const int nchunk = Dp >> 6;
    const int halfD = Dp >> 1;
    const int ntile = (T2 + TILE_ROWS - 1) / TILE_ROWS;
    const int nstep = ntile * nchunk;

    const float *arow = unitA + ((int64_t)g * T1 + (i < T1 ? i : T1 - 1)) * Dp + h * halfD;
    const float *bbase = unitB + (int64_t)g * T2 * Dp;

    float af[32];
    if (ONE_CHUNK) {
#pragma unroll
        for (int qd = 0; qd < 8; ++qd) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(arow + 4 * qd);
            af[4 * qd + 0] = v.x; af[4 * qd + 1] = v.y; af[4 * qd + 2] = v.z; af[4 * qd + 3] = v.w;
        }
    }

    // staging: a (tile, chunk) step is 32 rows x 64 floats = 512 float4; blockDim.x threads move
    // 512 / blockDim.x float4 each (W=8 -> 1, W=4 -> 2, W=2 -> 4, W=1 -> 8).
    const int per_thread = 512 / blockDim.x;
    f32x4 stage[8];
    auto stage_load = [&](int step) {
        const int jt = step / nchunk, c = step - jt * nchunk;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u < per_thread) {
                int f = u * blockDim.x + tid;       // float4 index inside the step: row*16 + q4
                int row = f >> 4, q4 = f & 15;      // q4 0..7 even-k half, 8..15 odd-k half
                int j = jt * TILE_ROWS + row;
                if (j >= T2) j = T2 - 1;
                const float *src = bbase + (int64_t)j * Dp + (q4 >> 3) * halfD + c * 32 + (q4 & 7) * 4;
                stage[u] = *reinterpret_cast<const f32x4 *>(src);
            }
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u < per_thread) {
                int f = u * blockDim.x + tid;
                int row = f >> 4, q4 = f & 15;
                *reinterpret_cast<f32x4 *>(&lds[buf][row * LDS_ROW + q4 * 4]) = stage[u];
            }
        }
    };

    float best = -INFINITY;
    int bidx = 0;
    f32x16 acc;

    stage_load(0);
    stage_write(0);
    __syncthreads();
    for (int step = 0; step < nstep; ++step) {
        const int jt = step / nchunk, c = step - jt * nchunk;
        if (step + 1 < nstep) stage_load(step + 1);
        if (!ONE_CHUNK) {
#pragma unroll
            for (int qd = 0; qd < 8; ++qd) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(arow + c * 32 + 4 * qd);
                af[4 * qd + 0] = v.x; af[4 * qd + 1] = v.y; af[4 * qd + 2] = v.z; af[4 * qd + 3] = v.w;
            }
        }
        if (c == 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
        }
        const float *brow = &lds[step & 1][col * LDS_ROW + h * 32];
        float bf[32];
#pragma unroll
        for (int qd = 0; qd < 8; ++qd) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(brow + 4 * qd);
            bf[4 * qd + 0] = v.x; bf[4 * qd + 1] = v.y; bf[4 * qd + 2] = v.z; bf[4 * qd + 3] = v.w;
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s], af[s], acc, 0, 0, 0);
        if (c == nchunk - 1) {
            const int jbase = jt * TILE_ROWS + 4 * h;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int j = jbase + (v & 3) + 8 * (v >> 2);
                const float sc = acc[v];
                bool ok = (j < T2) && !(distill_token && j == 0);
                if (ok && sc > best) {
                    best = sc;
                    bidx = j;
                }
            }
        }
        if (step + 1 < nstep) stage_write((step + 1) & 1);
        __syncthreads();
    }
    // the two lane halves hold the same A row, disjoint B rows: keep the larger, first index on ties
    float ob = __shfl_xor(best, 32);
    int oi = __shfl_xor(bidx, 32);
    if (ob > best || (ob == best && oi < bidx)) {
        best = ob;
        bidx = oi;
    }
    if (class_token && i == 0) {  // merge.py:59-60: the class token's row is all -inf
        best = -INFINITY;
        bidx = 0;
    }
    if (h == 0 && i < T1) {
        node_max[(int64_t)g * T1 + i] = best;
        node_idx[(int64_t)g * T1 + i] = bidx;
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
// -0 == +0, equal keys in ascending row order.  Each thread ranks one row against all T1 keys held
// in LDS (broadcast reads); the rank IS the position in edge_idx, so src/dst/unm are written directly.
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

__global__ __launch_bounds__(256) void k_rank_select(const float *__restrict__ node_max,
                                                     const int *__restrict__ node_idx, int n, int T1, int r,
                                                     int class_token, int distill_token,
                                                     int64_t *__restrict__ src_idx,
                                                     int64_t *__restrict__ dst_idx,
                                                     int64_t *__restrict__ unm_idx, int *__restrict__ rank_out,
                                                     int *__restrict__ row_map) {
    extern __shared__ __attribute__((aligned(16))) uint32_t keys[];
    const int g = blockIdx.y;
    const int T1p = (T1 + 3) & ~3;
    const float *nm = node_max + (int64_t)g * T1;
    for (int j = threadIdx.x; j < T1p; j += blockDim.x) keys[j] = (j < T1) ? sort_key(nm[j]) : 0u;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T1) return;
    const uint32_t ki = keys[i];
    int cnt = 0;
    const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
    for (int j4 = 0; j4 < (T1p >> 2); ++j4) {
        uint4 k = k4[j4];
        const int j = j4 << 2;
        cnt += (k.x > ki) || (k.x == ki && (j + 0) < i);
        cnt += (k.y > ki) || (k.y == ki && (j + 1) < i);
        cnt += (k.z > ki) || (k.z == ki && (j + 2) < i);
        cnt += (k.w > ki) || (k.w == ki && (j + 3) < i);
    }
    const int U = T1 - r;
    const int64_t gi = (int64_t)g * T1 + i;
    if (rank_out) rank_out[gi] = cnt;
    if (cnt < r) {
        const int d = node_idx[gi];
        src_idx[(int64_t)g * r + cnt] = i;
        dst_idx[(int64_t)g * r + cnt] = d;
        if (row_map) row_map[gi] = out_row_dst(d, U, distill_token);
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

template <typename TX, typename TS, int VEC, int OP>
__global__ __launch_bounds__(256) void k_merge_rows(const TX *__restrict__ x, const TS *__restrict__ size,
                                                    int n, int T_, int C, int r,
                                                    const int64_t *__restrict__ src_idx,
                                                    const int64_t *__restrict__ dst_idx,
                                                    const int64_t *__restrict__ unm_idx, int distill,
                                                    const uint8_t *__restrict__ keep, TX *__restrict__ xout,
                                                    TS *__restrict__ sout) {
    const int lane = threadIdx.x & 63;
    const int To = T_ - r;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= (int64_t)n * To) return;
    const int g = (int)(row / To);
    const int o = (int)(row - (int64_t)g * To);
    const int T1 = (T_ + 1) >> 1, U = T1 - r;

    // inverse of the output layout (merge.py:82-85)
    bool is_dst;
    int idx;
    if (!distill) {
        is_dst = o >= U;
        idx = is_dst ? o - U : o;
    } else if (o == 0) { is_dst = false; idx = 0; }
    else if (o == 1) { is_dst = true; idx = 0; }
    else if (o <= U) { is_dst = false; idx = o - 1; }
    else { is_dst = true; idx = o - U; }

    const TX *xg = x + (int64_t)g * T_ * C;
    const TS *sg = size ? size + (int64_t)g * T_ : nullptr;
    TX *orow = xout + row * C;
    const int64_t *srcg = src_idx ? src_idx + (int64_t)g * r : nullptr;
    const int64_t *dstg = dst_idx ? dst_idx + (int64_t)g * r : nullptr;

    if (!is_dst) {
        const int t = 2 * (int)unm_idx[(int64_t)g * U + idx];
        const TX *xr = xg + (int64_t)t * C;
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

    const int j = idx;
    const int t = 2 * j + 1;
    const TX *xr = xg + (int64_t)t * C;
    float s_own = 1.0f;
    if (OP == OP_WAVG) s_own = sg ? to_f32(sg[t]) : 1.0f;

    // edges into this destination, first 64 ranks (covers every r <= 64 in one ballot)
    unsigned long long mask0 = 0ull;
    bool kill = false;  // hybrid: some incoming edge is below the threshold (merge.py:326)
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
                        load_pack<TX, VEC>(xg + (int64_t)ts * C + c, v);
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
    if (OP == OP_WAVG && lane == 0) sout[row] = from_f32<TS>(ssum);
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
    } else if (!on && g_prof.on) {
        for (int i = 0; i < PROF_EVENTS; ++i) (void)hipEventDestroy(g_prof.ev[i]);
        g_prof.on = false;
        g_prof.valid = false;
    }
    return TOME_OK;
}

extern "C" int tome_profile_read(float *stage_ms, int max_stages) {
    if (!g_prof.on || !g_prof.valid || !stage_ms) return fail(TOME_EINVAL, "tome_profile_read: no profiled call");
    if (hipEventSynchronize(g_prof.ev[PROF_EVENTS - 1]) != hipSuccess)
        return fail(TOME_ELAUNCH, "tome_profile_read: event synchronize failed");
    for (int i = 0; i + 1 < PROF_EVENTS && i < max_stages; ++i)
        if (hipEventElapsedTime(&stage_ms[i], g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess)
            return fail(TOME_ELAUNCH, "tome_profile_read: elapsed time failed");
    return TOME_OK;
}

struct MatchWs {
    float *unitA, *unitB, *node_max;
    int *node_idx, *rank;
    size_t bytes;
};

static MatchWs carve(void *base, int64_t n, int64_t T, int64_t D) {
    const int64_t T1 = (T + 1) / 2, T2 = T / 2;
    const int64_t Dp = (D + 63) / 64 * 64;
    size_t off = 0;
    MatchWs w;
    char *b = (char *)base;
    w.unitA = (float *)(b + off); off = align_up(off + sizeof(float) * (size_t)(n * T1 * Dp), 256);
    w.unitB = (float *)(b + off); off = align_up(off + sizeof(float) * (size_t)(n * T2 * Dp), 256);
    w.node_max = (float *)(b + off); off = align_up(off + sizeof(float) * (size_t)(n * T1), 256);
    w.node_idx = (int *)(b + off); off = align_up(off + sizeof(int) * (size_t)(n * T1), 256);
    w.rank = (int *)(b + off); off = align_up(off + sizeof(int) * (size_t)(n * T1), 256);
    w.bytes = off;
    return w;
}

extern "C" size_t tome_match_workspace_bytes(int64_t n, int64_t T, int64_t D) {
    if (n <= 0 || T <= 0 || D <= 0) return 0;
    return carve(nullptr, n, T, D).bytes;
}

static int launch_select(const MatchWs &w, int64_t n, int64_t T, int64_t re, int class_token,
                         int distill_token, int64_t *src_idx, int64_t *dst_idx, int64_t *unm_idx,
                         float *node_max, int32_t *row_map, hipStream_t st) {
    const int T1 = (int)((T + 1) / 2);
    dim3 grid((T1 + 255) / 256, (unsigned)n);
    const size_t lds = sizeof(uint32_t) * (size_t)((T1 + 3) & ~3);
    hipLaunchKernelGGL(k_rank_select, grid, dim3(256), lds, st, w.node_max, w.node_idx, (int)n, T1, (int)re,
                       class_token, distill_token, src_idx, dst_idx, unm_idx, w.rank, row_map);
    if (int rc = check_launch("k_rank_select")) return rc;
    if (class_token) {
        hipLaunchKernelGGL(k_compact_unm, dim3((unsigned)n), dim3(256), 0, st, w.rank, T1, (int)re,
                           distill_token, unm_idx, row_map);
        if (int rc = check_launch("k_compact_unm")) return rc;
    }
    if (node_max) {
        hipError_t e = hipMemcpyAsync(node_max, w.node_max, sizeof(float) * (size_t)(n * T1),
                                      hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return fail(TOME_ELAUNCH, "node_max copy: %s", hipGetErrorString(e));
    }
    return TOME_OK;
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
    const int T1 = (int)((T + 1) / 2), T2 = (int)(T / 2);
    const int Dp = (int)((D + 63) / 64 * 64);

    // 1. unit vectors
    prof_mark(0, st);
    int tok = (int)(65536 / ((D + 1) * sizeof(float)));
    if (tok > 64) tok = 64;
    if (tok < 1) return fail(TOME_EINVAL, "tome_match: D=%lld too wide", (long long)D);
    const size_t lds1 = sizeof(float) * ((size_t)tok * (D + 1) + tok);
    const unsigned nb1 = (unsigned)((n * T + tok - 1) / tok);
    switch (dtype) {
    case TOME_F32:
        hipLaunchKernelGGL(k_unit_rows<float>, dim3(nb1), dim3(256), lds1, st, (const float *)metric, stride_n,
                           stride_t, (int)n, (int)T, (int)D, Dp, tok, w.unitA, w.unitB);
        break;
    case TOME_BF16:
        hipLaunchKernelGGL(k_unit_rows<bf16_t>, dim3(nb1), dim3(256), lds1, st, (const bf16_t *)metric, stride_n,
                           stride_t, (int)n, (int)T, (int)D, Dp, tok, w.unitA, w.unitB);
        break;
    case TOME_F16:
        hipLaunchKernelGGL(k_unit_rows<f16_t>, dim3(nb1), dim3(256), lds1, st, (const f16_t *)metric, stride_n,
                           stride_t, (int)n, (int)T, (int)D, Dp, tok, w.unitA, w.unitB);
        break;
    default: return fail(TOME_EINVAL, "tome_match: dtype %d", dtype);
    }
    if (int rc = check_launch("k_unit_rows")) return rc;
    prof_mark(1, st);

    // 2. similarity + row max/argmax
    const int tiles = (T1 + TILE_ROWS - 1) / TILE_ROWS;
    int W = tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1);
    const int wgpg = (tiles + W - 1) / W;
    const unsigned nb2 = (unsigned)(((n + 7) / 8) * 8 * wgpg);
    if (Dp == 64)
        hipLaunchKernelGGL(k_scores_rowmax<true>, dim3(nb2), dim3(64 * W), 0, st, w.unitA, w.unitB, (int)n, T1, T2,
                           Dp, wgpg, class_token, distill_token, w.node_max, w.node_idx);
    else
        hipLaunchKernelGGL(k_scores_rowmax<false>, dim3(nb2), dim3(64 * W), 0, st, w.unitA, w.unitB, (int)n, T1,
                           T2, Dp, wgpg, class_token, distill_token, w.node_max, w.node_idx);
    if (int rc = check_launch("k_scores_rowmax")) return rc;
    prof_mark(2, st);

    // 3. rank + select
    int rc = launch_select(w, n, T, re, class_token, distill_token, src_idx, dst_idx, unm_idx, node_max, row_map, st);
    prof_mark(3, st);
    g_prof.valid = g_prof.on && rc == TOME_OK;
    return rc;
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
                       distill_token, w.node_max, w.node_idx);
    if (int rc = check_launch("k_rowmax_given")) return rc;
    return launch_select(w, n, T, re, class_token, distill_token, src_idx, dst_idx, unm_idx, node_max, row_map, st);
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

template <typename TX, typename TS, int OP>
static int launch_merge_rows(const void *x, const void *size, int64_t n, int64_t T, int64_t C, int64_t r,
                             const int64_t *src, const int64_t *dst, const int64_t *unm, int distill,
                             const uint8_t *keep, void *xout, void *sout, hipStream_t st) {
    constexpr int VEC = 16 / sizeof(TX);
    const int64_t rows = n * (T - r);
    const unsigned nb = (unsigned)((rows + 3) / 4);
    const bool vec_ok = (C % VEC == 0) && aligned16(x) && aligned16(xout);
    if (vec_ok)
        hipLaunchKernelGGL((k_merge_rows<TX, TS, VEC, OP>), dim3(nb), dim3(256), 0, st, (const TX *)x,
                           (const TS *)size, (int)n, (int)T, (int)C, (int)r, src, dst, unm, distill, keep, (TX *)xout,
                           (TS *)sout);
    else
        hipLaunchKernelGGL((k_merge_rows<TX, TS, 1, OP>), dim3(nb), dim3(256), 0, st, (const TX *)x,
                           (const TS *)size, (int)n, (int)T, (int)C, (int)r, src, dst, unm, distill, keep, (TX *)xout,
                           (TS *)sout);
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
