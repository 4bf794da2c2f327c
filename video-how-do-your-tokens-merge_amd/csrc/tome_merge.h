// tome_merge.h -- part of the single translation unit csrc/tome_kernels.hip (merge / unmerge row kernels).
#pragma once
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
    if (L.inner == 1) return p + L.base + (int64_t)g * L.outer_stride;  // (no integer division on the plain layout)
    return p + L.base + (int64_t)(g / L.inner) * L.outer_stride + (int64_t)(g % L.inner) * L.inner_stride;
}

// Optional LayerNorm fused behind the merge (the block's norm2 -- tome/patch/videomae.py:27 `self.norm2(x)`,
// vivit.py:41 `layernorm_after`): y = (x' - mean) * rstd * weight + bias over the channels of every merged
// row, fp32 statistics (two passes over the registers: mean, then centred variance), written next to x'.
struct LnArgs {
    const void *weight, *bias;  // [C] of the token dtype
    void *y;                    // [n, T-r, C] of the token dtype, same row layout as x_out
    float eps;
    const void *addend;         // optional [n, T, C]: the tokens that are merged are round(x + addend), i.e. the
                                // block's residual `x = x + attn(...)` (videomae.py:20,25) is taken on the fly
    // The addend may live in a layout of its own (a_own): TimeSformer's second residual is the spatial attention's
    // output [(b t), 1 + p, m] (timesformer.py:40-52), read where it lies instead of being permuted to 'b (p t) m'
    // first; the class rows' addend (the class token averaged over the frames, timesformer.py:42-44) is then a
    // separate [B, C] tensor.
    int a_own;
    TokLayout la;
    const void *cls_addend;
    int y_group;  // k_add_ln_rows only: see there
    // optional [C]: the rows of x_out are stored as round(x' + xbias) while y stays LayerNorm(x').  The caller's next
    // GEMM accumulates onto x_out in place (`x = x + fc2(...)`, tome/patch/videomae.py:29, as beta = 1 on the residual
    // buffer), so its bias has to be in the buffer beforehand and the separate residual add disappears.
    const void *xbias;
};

// round(v + xbias[c..]) for a 16-byte pack of stored tokens
template <typename TX, int VEC>
__device__ __forceinline__ uint4 add_xbias(const uint4 &v, const TX *__restrict__ xb) {
    Pack<TX, VEC> pk;
    __builtin_memcpy(&pk, &v, 16);
    float b8[VEC];
    load_pack<TX, VEC>(xb, b8);
#pragma unroll
    for (int e = 0; e < VEC; ++e) pk.e[e] = from_f32<TX>(__fadd_rn(to_f32(pk.e[e]), b8[e]));
    uint4 o;
    __builtin_memcpy(&o, &pk, 16);
    return o;
}

// 16-byte moves of the streaming kernels (k_merge_rows_fast, k_add_ln_rows, k_add_ln_regroup): tokens are read once
// and written once, 0.3-1.2 GB per launch against 256 MB of Infinity Cache, so both directions are issued
// non-temporal (TOME_NT bit 0 loads, bit 1 stores).  Measured at batch 128, same box, alternating builds: merge+LN
// 226 -> 214 us (5.12 -> 5.40 TB/s), add+LN 215 -> 189 us (5.37 -> 6.12 TB/s); inside the model (rocprofv3) 228 -> 221
// and 204 -> 188 us with the GEMMs that read the results unchanged (581 vs 578 us).
#ifndef TOME_NT
#define TOME_NT 3
#endif
typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16(const void *p) {
    if (TOME_NT & 1) {
        const nt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(p));
        return uint4{v.x, v.y, v.z, v.w};
    }
    return *reinterpret_cast<const uint4 *>(p);
}
__device__ __forceinline__ void st16(void *p, const uint4 &v) {
    if (TOME_NT & 2) {
        nt_u32x4 w;
        w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
        __builtin_nontemporal_store(w, reinterpret_cast<nt_u32x4 *>(p));
    } else {
        *reinterpret_cast<uint4 *>(p) = v;
    }
}

// round(x + a) in the token dtype, element-wise on two 16-byte packs (what torch's `x + a` stores)
template <typename TX, int VEC>
__device__ __forceinline__ Pack<TX, VEC> add_packs(const Pack<TX, VEC> &x, const Pack<TX, VEC> &a) {
    Pack<TX, VEC> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.e[e] = from_f32<TX>(__fadd_rn(to_f32(x.e[e]), to_f32(a.e[e])));
    return o;
}

// size' of one output row, and -- for the proportional-attention bias of the next block
// (`size.log()`, tome/patch/videomae.py:62-63, timesformer.py:73-74, motionformer.py:107-111, vivit.py:103-104)
// -- log of the STORED size, fp32 logf rounded to the size dtype: what torch's `size.log()` gives on this device.
template <typename TS>
__device__ __forceinline__ void store_size(TS *__restrict__ srow, TS *__restrict__ lrow, float s) {
    const TS st = from_f32<TS>(s);
    *srow = st;
    if (lrow) *lrow = from_f32<TS>(logf(to_f32(st)));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm tail of the streaming kernels (k_merge_rows_fast, k_add_ln_rows, k_add_ln_regroup).  These kernels sit at
// the vector-issue limit of the SIMDs, not only at the HBM limit (measured in round 2: adding ~75 vector instructions
// per lane cost the merge kernel 6 %; occupancy and index prefetch changed nothing), so the tail is written for few
// instructions:
//   * the sum of a 16-byte chunk by dot products with a vector of ones (v_dot2c_f32_bf16 / _f16: two elements per
//     instruction, no unpacking);
//   * wave totals by a DPP scan (row_shr 1/2/4/8, row_bcast 15/31) read back from lane 63 as a wave-uniform scalar,
//     instead of six ds_bpermute round trips per total -- and only for the rows the wave really holds (R of 4);
//   * the centred values d = x - mean kept in registers between the variance pass and the output pass;
//   * y = d * (rstd * w) + b as one multiply and one fma per element; 1/C as a multiplication, rstd by v_rsq_f32.
// Two passes as before (mean, then centred variance): a row far from zero keeps its variance.
// ------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_total(float v) {  // sum over the 64 lanes, wave-uniform
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of every row of 16 holds the row's total
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

typedef __bf16 dot_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 dot_f16x2 __attribute__((ext_vector_type(2)));
template <typename TX> __device__ __forceinline__ float chunk_sum(const uint4 &v);
template <> __device__ __forceinline__ float chunk_sum<bf16_t>(const uint4 &v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w}, ones = 0x3f803f80u;
    dot_bf16x2 o, p;
    __builtin_memcpy(&o, &ones, 4);
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_memcpy(&p, &w[i], 4);
        t = __builtin_amdgcn_fdot2_f32_bf16(p, o, t, false);
    }
    return t;
}
template <> __device__ __forceinline__ float chunk_sum<f16_t>(const uint4 &v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w}, ones = 0x3c003c00u;
    dot_f16x2 o, p;
    __builtin_memcpy(&o, &ones, 4);
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_memcpy(&p, &w[i], 4);
        t = __builtin_amdgcn_fdot2(p, o, t, false);
    }
    return t;
}

__device__ __forceinline__ float pick4(int rr, float a0, float a1, float a2, float a3, int R) {
    float v = a0;
    if (R > 1) v = rr == 1 ? a1 : v;
    if (R > 2) {
        v = rr == 2 ? a2 : v;
        v = rr == 3 ? a3 : v;
    }
    return v;
}

// raw[it]: the stored bits of chunk `it` of this lane (16-bit tokens, 8 per chunk), rowof[it] its row 0..R-1 or -1;
// store_y(it, rr, cc, bits) writes one normalised chunk (cc = 16-byte column of the chunk in its row).
template <typename TX, int NIT, typename StoreY>
__device__ __forceinline__ void ln_rows(const uint4 (&raw)[NIT], const int (&rowof)[NIT], int R, int cpr, int C,
                                        float eps, const TX *__restrict__ lw, const TX *__restrict__ lb, int lane,
                                        StoreY store_y) {
    constexpr int VEC = 8;
    const float inv_c = __builtin_amdgcn_rcpf((float)C);
    // weight and bias chunks of every chunk column this lane holds: requested now, used after the two reductions
    uint4 wraw[NIT], braw[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rr = rowof[it];
        const int cc = rr < 0 ? 0 : it * WAVE + lane - rr * cpr;
        wraw[it] = *(reinterpret_cast<const uint4 *>(lw) + cc);
        braw[it] = *(reinterpret_cast<const uint4 *>(lb) + cc);
    }
    float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rr = rowof[it];
        const float t = rr >= 0 ? chunk_sum<TX>(raw[it]) : 0.0f;
        p0 += rr == 0 ? t : 0.0f;
        if (R > 1) p1 += rr == 1 ? t : 0.0f;
        if (R > 2) {
            p2 += rr == 2 ? t : 0.0f;
            p3 += rr == 3 ? t : 0.0f;
        }
    }
    float m0 = wave_total(p0) * inv_c, m1 = 0.0f, m2 = 0.0f, m3 = 0.0f;
    if (R > 1) m1 = wave_total(p1) * inv_c;
    if (R > 2) {
        m2 = wave_total(p2) * inv_c;
        m3 = wave_total(p3) * inv_c;
    }
    float d[NIT][VEC];
    p0 = p1 = p2 = p3 = 0.0f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rr = rowof[it];
        const float m = pick4(rr, m0, m1, m2, m3, R);
        Pack<TX, VEC> pk;
        __builtin_memcpy(&pk, &raw[it], 16);
        float u = 0.0f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            d[it][e] = to_f32(pk.e[e]) - m;
            u = __fmaf_rn(d[it][e], d[it][e], u);
        }
        u = rr >= 0 ? u : 0.0f;
        p0 += rr == 0 ? u : 0.0f;
        if (R > 1) p1 += rr == 1 ? u : 0.0f;
        if (R > 2) {
            p2 += rr == 2 ? u : 0.0f;
            p3 += rr == 3 ? u : 0.0f;
        }
    }
    float r0 = __builtin_amdgcn_rsqf(wave_total(p0) * inv_c + eps), r1 = 0.0f, r2 = 0.0f, r3 = 0.0f;
    if (R > 1) r1 = __builtin_amdgcn_rsqf(wave_total(p1) * inv_c + eps);
    if (R > 2) {
        r2 = __builtin_amdgcn_rsqf(wave_total(p2) * inv_c + eps);
        r3 = __builtin_amdgcn_rsqf(wave_total(p3) * inv_c + eps);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rr = rowof[it];
        if (rr < 0) continue;
        const int cc = it * WAVE + lane - rr * cpr;
        const float rs = pick4(rr, r0, r1, r2, r3, R);
        Pack<TX, VEC> pw, pb, pk;
        __builtin_memcpy(&pw, &wraw[it], 16);
        __builtin_memcpy(&pb, &braw[it], 16);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            pk.e[e] = from_f32<TX>(__fmaf_rn(d[it][e], to_f32(pw.e[e]) * rs, to_f32(pb.e[e])));
        uint4 yv;
        __builtin_memcpy(&yv, &pk, 16);
        store_y(it, rr, cc, yv);
    }
}

// The tail of one destination row held by a whole wave as two 16-byte column chunks per lane (acc0 at column c0, acc1
// at c1; a0 / a1 say which exist): division by the summed size (or the count, for "mean"), the stores of x' (with the
// folded bias), size' (and its log), and the fused LayerNorm of the row as stored.  Shared by merge_dst_row and
// merge_dst_row_r64, so both build bit-identical rows.
// (pre: the lane's chunks of xbias / LayerNorm weight / bias when the caller has requested them ahead of its own
// waits -- merge_dst_row_r64; nullptr: fetched here)
template <typename TX, int VEC> struct RowConsts { Pack<TX, VEC> xb0, xb1, w0, w1, b0, b1; };

template <typename TX, int VEC>
__device__ __forceinline__ void unpack_f32(const Pack<TX, VEC> &p, float (&out)[VEC]) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) out[e] = to_f32(p.e[e]);
}

template <typename TX, typename TS, int VEC, int OP, bool LN, bool PRE = false>
__device__ __forceinline__ void dst_row_finish(float (&acc0)[VEC], float (&acc1)[VEC], bool a0, bool a1, int c0, int c1,
                                               float ssum, int cnt, int C, TX *__restrict__ orow, TS *__restrict__ srow,
                                               TS *__restrict__ lrow, const LnArgs *ln, TX *__restrict__ yrow, int lane,
                                               const RowConsts<TX, VEC> pre = RowConsts<TX, VEC>()) {
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
    if (LN && ln->xbias) {
        const TX *xb = reinterpret_cast<const TX *>(ln->xbias);
        float b8[VEC], o8[VEC];
        if (a0) {
            if (PRE) unpack_f32<TX, VEC>(pre.xb0, b8); else load_pack<TX, VEC>(xb + c0, b8);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o8[e] = __fadd_rn(to_f32(from_f32<TX>(acc0[e])), b8[e]);
            store_pack<TX, VEC>(orow + c0, o8);
        }
        if (a1) {
            if (PRE) unpack_f32<TX, VEC>(pre.xb1, b8); else load_pack<TX, VEC>(xb + c1, b8);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o8[e] = __fadd_rn(to_f32(from_f32<TX>(acc1[e])), b8[e]);
            store_pack<TX, VEC>(orow + c1, o8);
        }
    } else {
        if (a0) store_pack<TX, VEC>(orow + c0, acc0);
        if (a1) store_pack<TX, VEC>(orow + c1, acc1);
    }
    if (OP == OP_WAVG && lane == 0) store_size<TS>(srow, lrow, ssum);
    if (LN) {
        // LayerNorm of the row as stored (rounded to the token dtype), the whole row is in this wave
        float part = 0.0f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            acc0[e] = a0 ? to_f32(from_f32<TX>(acc0[e])) : 0.0f;
            acc1[e] = a1 ? to_f32(from_f32<TX>(acc1[e])) : 0.0f;
            part += acc0[e] + acc1[e];
        }
        const float mean = wave_sum(part) / (float)C;
        part = 0.0f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float d0 = a0 ? acc0[e] - mean : 0.0f, d1 = a1 ? acc1[e] - mean : 0.0f;
            part += d0 * d0 + d1 * d1;
        }
        const float rstd = 1.0f / __builtin_sqrtf(wave_sum(part) / (float)C + ln->eps);
        const TX *lw = reinterpret_cast<const TX *>(ln->weight), *lb = reinterpret_cast<const TX *>(ln->bias);
        if (a0) {
            float w8[VEC], b8[VEC];
            if (PRE) { unpack_f32<TX, VEC>(pre.w0, w8); unpack_f32<TX, VEC>(pre.b0, b8); }
            else { load_pack<TX, VEC>(lw + c0, w8); load_pack<TX, VEC>(lb + c0, b8); }
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc0[e] = (acc0[e] - mean) * rstd * w8[e] + b8[e];
            store_pack<TX, VEC>(yrow + c0, acc0);
        }
        if (a1) {
            float w8[VEC], b8[VEC];
            if (PRE) { unpack_f32<TX, VEC>(pre.w1, w8); unpack_f32<TX, VEC>(pre.b1, b8); }
            else { load_pack<TX, VEC>(lw + c1, w8); load_pack<TX, VEC>(lb + c1, b8); }
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc1[e] = (acc1[e] - mean) * rstd * w8[e] + b8[e];
            store_pack<TX, VEC>(yrow + c1, acc1);
        }
    }
}

// merge_dst_row for the launches whose whole index list fits one lane each (r <= 64) and whose rows are at most two
// 16-byte chunks per lane: the form the edge waves of TimeSformer / Motionformer's frame groups run (196 tokens, r up to
// 32: a third to all of the destinations receive sources, so this path IS the kernel there).  merge_dst_row walks five
// dependent memory round trips (dst_idx[k] -> own row | dst_idx scan -> src_idx -> source sizes -> source rows); here
// lane L holds dst_idx[L], src_idx[L] and the keep flag of rank L from ONE batch of loads, every decision (is k the first
// edge into its destination, which ranks merge into it, in which order) is a ballot over registers, and the second
// batch holds everything else at once: the own row, the first NB source rows (with their residual rows), all sizes, the
// bias / LayerNorm constants.  Two round trips.  Arithmetic and order are merge_dst_row's (own term, then the sources
// in rank = lane order; shared tail dst_row_finish), so the rows are bit-identical.
// Returns without doing anything when rank k is not the first edge into its destination.
template <typename TX, typename TS, int VEC, int OP, bool LN>
__device__ __forceinline__ void merge_dst_row_r64(const TX *__restrict__ xg, const TS *__restrict__ sg, int C,
                                                  int64_t tstride, int r, int g, int k,
                                                  const int64_t *__restrict__ srcg, const int64_t *__restrict__ dstg,
                                                  const uint8_t *__restrict__ keep, TX *__restrict__ og,
                                                  int64_t ostride, TS *__restrict__ sog, TS *__restrict__ log, int U,
                                                  int distill, int lane, const LnArgs *ln, TX *__restrict__ yg,
                                                  const TX *__restrict__ ag, int64_t astride) {
    typedef Pack<TX, VEC> __attribute__((aligned(sizeof(TX) * VEC))) PK;
#ifndef TOME_R64_NB
#define TOME_R64_NB 1
#endif
    constexpr int NB = TOME_R64_NB;  // source rows requested together with the own row
    // ---- batch 1: the group's index lists, one rank per lane
    const int li = lane < r ? lane : 0;
    const int d_l = lane < r ? (int)dstg[li] : -1;
    const int s_l = (int)srcg[li];
    const int keep_l = keep ? (int)keep[(int64_t)g * r + li] : 1;
    const int j = __builtin_amdgcn_readlane(d_l, k);
    const unsigned long long mk = __ballot(d_l == j);  // the ranks that merge into j, rank order = bit order
    if ((int)__ffsll((long long)mk) - 1 != k) return;  // an earlier rank builds this destination
    const bool kill = keep && OP != OP_DROP && (__ballot((d_l == j) && keep_l == 0) != 0ull);
    // ---- batch 2: everything the row needs
    const int t = 2 * j + 1;
    const bool has_s = (OP == OP_WAVG) && sg;
    const TS sz_raw = (has_s ? sg : reinterpret_cast<const TS *>(xg))[has_s ? 2 * s_l : 0];  // size of rank `lane`'s source
    const TS own_raw = (has_s ? sg : reinterpret_cast<const TS *>(xg))[has_s ? t : 0];
    const int c0 = lane * VEC, c1 = (WAVE + lane) * VEC;
    const bool a0 = c0 < C, a1 = c1 < C;
    const int c0s = a0 ? c0 : 0, c1s = a1 ? c1 : 0;  // (loads are unconditional: a lane without a chunk re-reads column 0)
    const bool add = LN && ag;
    const TX *xr = xg + (int64_t)t * tstride;
    const TX *ar = add ? ag + (int64_t)t * astride : xr;
    const PK x0 = *reinterpret_cast<const PK *>(xr + c0s), x1 = *reinterpret_cast<const PK *>(xr + c1s);
    PK y0 = x0, y1 = x1;
    if (add) { y0 = *reinterpret_cast<const PK *>(ar + c0s); y1 = *reinterpret_cast<const PK *>(ar + c1s); }
    unsigned long long rest = mk;
    int bsel[NB];
    bool have[NB];
    PK p0[NB], p1[NB], q0[NB], q1[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        have[u] = rest != 0ull;
        bsel[u] = have[u] ? (int)__ffsll((long long)rest) - 1 : k;
        rest = have[u] ? (rest & (rest - 1ull)) : rest;
        const int tsu = 2 * __builtin_amdgcn_readlane(s_l, bsel[u]);
        const TX *sr = xg + (int64_t)tsu * tstride;
        p0[u] = *reinterpret_cast<const PK *>(sr + c0s);
        p1[u] = *reinterpret_cast<const PK *>(sr + c1s);
        if (add) {
            const TX *sa = ag + (int64_t)tsu * astride;
            q0[u] = *reinterpret_cast<const PK *>(sa + c0s);
            q1[u] = *reinterpret_cast<const PK *>(sa + c1s);
        }
    }
    RowConsts<TX, VEC> rc;
    if (LN) {
        // (every field is loaded, the bias chunks from the weight when there is no folded bias: no conditional
        // initialisation, so the struct lives in registers)
        const TX *lw = reinterpret_cast<const TX *>(ln->weight), *lb = reinterpret_cast<const TX *>(ln->bias);
        const TX *xb = ln->xbias ? reinterpret_cast<const TX *>(ln->xbias) : lw;
        rc.w0 = *reinterpret_cast<const PK *>(lw + c0s);
        rc.w1 = *reinterpret_cast<const PK *>(lw + c1s);
        rc.b0 = *reinterpret_cast<const PK *>(lb + c0s);
        rc.b1 = *reinterpret_cast<const PK *>(lb + c1s);
        rc.xb0 = *reinterpret_cast<const PK *>(xb + c0s);
        rc.xb1 = *reinterpret_cast<const PK *>(xb + c1s);
    }
    // ---- arithmetic, merge_dst_row's order
    const float sz_l = has_s ? to_f32(sz_raw) : 1.0f;
    const float s_own = has_s ? to_f32(own_raw) : 1.0f;
    float acc0[VEC], acc1[VEC];
    {
        const Pack<TX, VEC> s0 = add ? add_packs<TX, VEC>(x0, y0) : (Pack<TX, VEC>)x0;
        const Pack<TX, VEC> s1 = add ? add_packs<TX, VEC>(x1, y1) : (Pack<TX, VEC>)x1;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            acc0[e] = to_f32(s0.e[e]);
            acc1[e] = to_f32(s1.e[e]);
        }
    }
    float ssum = s_own;
    int cnt = 1;
    if (OP == OP_WAVG) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            acc0[e] = __fmul_rn(acc0[e], s_own);
            acc1[e] = __fmul_rn(acc1[e], s_own);
        }
    }
    if (kill) {
        if (OP == OP_WAVG) ssum = __fmul_rn(ssum, 0.0f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            acc0[e] = __fmul_rn(acc0[e], 0.0f);
            acc1[e] = __fmul_rn(acc1[e], 0.0f);
        }
    }
    for (;;) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (!have[u]) continue;  // (wave-uniform)
            const float sq = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sz_l), bsel[u]));
            const Pack<TX, VEC> v0 = add ? add_packs<TX, VEC>(p0[u], q0[u]) : (Pack<TX, VEC>)p0[u];
            const Pack<TX, VEC> v1 = add ? add_packs<TX, VEC>(p1[u], q1[u]) : (Pack<TX, VEC>)p1[u];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float f0 = to_f32(v0.e[e]), f1 = to_f32(v1.e[e]);
                acc0[e] = reduce_step<OP>(acc0[e], (OP == OP_WAVG) ? __fmul_rn(f0, sq) : f0);
                acc1[e] = reduce_step<OP>(acc1[e], (OP == OP_WAVG) ? __fmul_rn(f1, sq) : f1);
            }
            if (OP == OP_WAVG) ssum = __fadd_rn(ssum, sq);
            ++cnt;
        }
        if (rest == 0ull) break;
        // a destination with more than NB sources (rare): the next NB, one more round trip
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            have[u] = rest != 0ull;
            bsel[u] = have[u] ? (int)__ffsll((long long)rest) - 1 : k;
            rest = have[u] ? (rest & (rest - 1ull)) : rest;
            const int tsu = 2 * __builtin_amdgcn_readlane(s_l, bsel[u]);
            const TX *sr = xg + (int64_t)tsu * tstride;
            p0[u] = *reinterpret_cast<const PK *>(sr + c0s);
            p1[u] = *reinterpret_cast<const PK *>(sr + c1s);
            if (add) {
                const TX *sa = ag + (int64_t)tsu * astride;
                q0[u] = *reinterpret_cast<const PK *>(sa + c0s);
                q1[u] = *reinterpret_cast<const PK *>(sa + c1s);
            }
        }
    }
    const int o = out_row_dst(j, U, distill);
    dst_row_finish<TX, TS, VEC, OP, LN, LN>(acc0, acc1, a0, a1, c0, c1, ssum, cnt, C, og + (int64_t)o * ostride,
                                            sog ? sog + o : nullptr, log ? log + o : nullptr, ln,
                                            LN ? yg + (int64_t)o * ostride : nullptr, lane, rc);
}

// One destination row (odd token 2j+1 plus every source merged into it), whole wave, contract order:
// own term first, then the sources in src_idx (rank) order found by ballot-scanning dst_idx.
// Two shapes of the same arithmetic:
//   * rows of at most 2*64 lane-chunks (C <= 1024 bf16 / 512 fp32 with 16-byte lanes): the sources of one
//     64-rank block are compacted onto lanes 0..nsrc-1 (ds_permute), their tokens and sizes fetched as ONE
//     vector load each, and their row chunks fetched four sources at a time before the sequential adds --
//     a destination with k sources costs ~2 + k/4 dependent memory round trips instead of ~2k;
//   * anything wider: the plain sequential loop.
template <typename TX, typename TS, int VEC, int OP, bool LN = false>
__device__ __forceinline__ void merge_dst_row(const TX *__restrict__ xg, const TS *__restrict__ sg, int C,
                                              int64_t tstride, int r, int g, int j,
                                              const int64_t *__restrict__ srcg,
                                              const int64_t *__restrict__ dstg, const uint8_t *__restrict__ keep,
                                              TX *__restrict__ orow, TS *__restrict__ srow, int lane,
                                              const LnArgs *ln = nullptr, TX *__restrict__ yrow = nullptr,
                                              const TX *__restrict__ ag = nullptr, TS *__restrict__ lrow = nullptr,
                                              int64_t astride = 0) {
    const int t = 2 * j + 1;
    const TX *xr = xg + (int64_t)t * tstride;
    float s_own = 1.0f;
    if (OP == OP_WAVG) s_own = sg ? to_f32(sg[t]) : 1.0f;
    const bool narrow_row = C <= 2 * WAVE * VEC;
    constexpr int NB = VEC >= 8 ? 2 : 4;  // source rows fetched together (register budget)

    if (narrow_row) {
        const int c0 = lane * VEC, c1 = (WAVE + lane) * VEC;
        const bool a0 = c0 < C, a1 = c1 < C;
        typedef Pack<TX, VEC> __attribute__((aligned(sizeof(TX) * VEC))) PKA;
        float acc0[VEC], acc1[VEC];
        if (LN && ag) {
            const TX *ar = ag + (int64_t)t * astride;
            PKA x0, x1, y0, y1;
            if (a0) { x0 = *reinterpret_cast<const PKA *>(xr + c0); y0 = *reinterpret_cast<const PKA *>(ar + c0); }
            if (a1) { x1 = *reinterpret_cast<const PKA *>(xr + c1); y1 = *reinterpret_cast<const PKA *>(ar + c1); }
            if (a0) { const Pack<TX, VEC> s0 = add_packs<TX, VEC>(x0, y0);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc0[e] = to_f32(s0.e[e]); }
            if (a1) { const Pack<TX, VEC> s1 = add_packs<TX, VEC>(x1, y1);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc1[e] = to_f32(s1.e[e]); }
        } else {
            if (a0) load_pack<TX, VEC>(xr + c0, acc0);
            if (a1) load_pack<TX, VEC>(xr + c1, acc1);
        }
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
                    PK q0[NB], q1[NB];  // ... and of the addend rows when the residual is fused
                    float sq[NB];
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const int e = (e0 + u < nsrc) ? e0 + u : nsrc - 1;
                        const int tsu = __builtin_amdgcn_readlane(tsL, e);
                        sq[u] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sL), e));
                        const TX *sr = xg + (int64_t)tsu * tstride;
                        if (a0) p0[u] = *reinterpret_cast<const PK *>(sr + c0);
                        if (a1) p1[u] = *reinterpret_cast<const PK *>(sr + c1);
                        if (LN && ag) {
                            const TX *sa = ag + (int64_t)tsu * astride;
                            if (a0) q0[u] = *reinterpret_cast<const PK *>(sa + c0);
                            if (a1) q1[u] = *reinterpret_cast<const PK *>(sa + c1);
                        }
                    }
                    if (LN && ag) {
#pragma unroll
                        for (int u = 0; u < NB; ++u) {
                            if (a0) p0[u] = add_packs<TX, VEC>(p0[u], q0[u]);
                            if (a1) p1[u] = add_packs<TX, VEC>(p1[u], q1[u]);
                        }
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
        dst_row_finish<TX, TS, VEC, OP, LN>(acc0, acc1, a0, a1, c0, c1, ssum, cnt, C, orow, srow, lrow, ln, yrow, lane);
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
    if (OP == OP_WAVG && lane == 0) store_size<TS>(srow, lrow, ssum);
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
                                                    TS *__restrict__ sout, TokLayout lin, TokLayout lout,
                                                    TS *__restrict__ lsout) {
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
        if (OP == OP_WAVG && lane == 0) store_size<TS>(sout + row, lsout ? lsout + row : nullptr, s);
        return;
    }
    merge_dst_row<TX, TS, VEC, OP>(xg, sg, C, lin.tok_stride, r, g, idx, src_idx ? src_idx + (int64_t)g * r : nullptr,
                                   dst_idx ? dst_idx + (int64_t)g * r : nullptr, keep, orow,
                                   sout ? sout + row : nullptr, lane, nullptr, nullptr, nullptr,
                                   lsout ? lsout + row : nullptr);
}

// ------------------------------------------------------------------------------------------------
// k_merge_rows_fast: the HBM-bound form for rows made of whole 16-byte chunks (C*sizeof(TX) % 16 == 0,
// at most 384 chunks per R rows).  A wave owns R consecutive OUTPUT rows of one group, flattens their
// R*cpr 16-byte chunks over its lanes (6 chunks per lane, every lane busy, stores contiguous across the
// R rows) and issues all of its loads before touching any of them, so ~6 KiB per wave are in flight.
// Rows that nothing merges into are moved as raw bits when their size is 1 ((x*1)/1 == x bit for bit)
// or scaled in fp32 otherwise; the few rows that receive sources are finished by merge_dst_row.
// How the loads stay together (checked in the ISA, tools/kernel_usage.py --keep): they are UNCONDITIONAL -- a lane
// without a chunk re-reads the start of the wave's first row -- because hipcc sinks the first use of a conditionally
// loaded value into the `if`, with an s_waitcnt vmcnt(0) behind every load; the size is read unconditionally for the
// same reason; which rows receive sources is decided after the row loads are out.  Launch geometry: grid.x = the
// blocks of one group (streaming waves, then r edge waves), grid.(y, z) = the group, class-token rows behind the
// groups -- no wave starts with an integer division.
// ------------------------------------------------------------------------------------------------
#define FAST_NIT 6
#define FAST_MAXR 4

// How the hardware workgroups of a k_merge_rows_fast launch map onto (group, block of the group).  on = 0: grid
// (blocks of one group, group.y, group.z) as launched.  on = 1: a 1-D grid of 8 * per_xcd workgroups renumbered so that
// each of the 8 XCDs (workgroup L -> XCD L % 8) walks whole groups in order.
struct MergeSched {
    unsigned bpg, total, per_xcd;  // blocks per group, bpg * (groups + class-row block rows), ceil(total / 8)
    int on;
    unsigned long long magic;      // ceil(2^40 / bpg)
};

// EAGER (round 3): when r is a large share of the destinations (TimeSformer / Motionformer late layers, r = 32 sweeps:
// r >= T2 / 4) the streaming waves first wait for dst_idx and do NOT read the rows that receive sources -- those are
// read (own row + sources) by the edge waves anyway; without it they were read twice (measured, tools/regroup_layers.py:
// 5.4 TB/s at 196 -> 164 tokens per group falling to 3.2 at 68 -> 36).  Costs the streaming waves of such launches one
// dependent round trip (index -> rows), which the default path avoids for the r << T2 case of the benchmark.
template <typename TX, typename TS, int OP, int NIT, bool LN = false, bool EAGER = false>
__global__ __launch_bounds__(256) void k_merge_rows_fast(const TX *__restrict__ x, const TS *__restrict__ size,
                                                         int n, int T_, int C, int r, int R, int cpr,
                                                         int rg_per_group, const int64_t *__restrict__ src_idx,
                                                         const int64_t *__restrict__ dst_idx,
                                                         const int64_t *__restrict__ unm_idx, int distill,
                                                         const uint8_t *__restrict__ keep, TX *__restrict__ xout,
                                                         TS *__restrict__ sout, TokLayout lin, TokLayout lout,
                                                         int cls_rows, LnArgs ln, TS *__restrict__ lsout,
                                                         MergeSched sch) {
    constexpr int VEC = 16 / sizeof(TX);
    const int lane = threadIdx.x & 63;
    const int To = T_ - r;
    // Grid: x = the blocks of ONE merge group (four waves each), (y, z) = the group -- so that no wave starts with
    // integer divisions of a flat index (they were ~250 dependent scalar instructions before the first load).  Waves
    // 0 .. rg-1 of a group stream R output rows each, the next r waves are the edge waves, the rest leave; the class
    // tokens kept aside by the regrouped callers are handled by the blocks of the (y, z) rows behind the n groups.
    // (rg_per_group = ceil((T - r) / R), from the host)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int gy, bx;
    if (sch.on) {
        // XCD-aware numbering (MergeSched): hardware workgroup L runs on XCD L % 8; virtual block V walks the groups in
        // order WITHIN one XCD, so an XCD works on whole groups (their index lists stay in its L2) and every XCD sees the
        // same mix of streaming and edge blocks at any time, whatever gridDim.x is
        const unsigned L = blockIdx.x;
        const unsigned V = (L & 7u) * sch.per_xcd + (L >> 3);
        if (V >= sch.total) return;
        const unsigned q = (unsigned)(((unsigned long long)V * sch.magic) >> 40);  // V / bpg (exact: V * bpg < 2^40)
        gy = (int)q;
        bx = (int)(V - q * sch.bpg);
    } else {
        gy = (int)(blockIdx.z * gridDim.y + blockIdx.y);
        bx = (int)blockIdx.x;
    }
    const int lw = bx * (int)(blockDim.x >> 6) + wv;
    if (gy >= n) {
        // the class tokens kept aside by the regrouped callers (timesformer.py:89,107): plain row copies
        const int64_t b = ((int64_t)(gy - n) * sch.bpg + bx) * (blockDim.x >> 6) + wv;
        if (b < cls_rows) {
            const uint4 *src = reinterpret_cast<const uint4 *>(x + b * lin.outer_stride);
            uint4 *dst = reinterpret_cast<uint4 *>(xout + b * lout.outer_stride);
            if (!LN) {
                for (int c = lane; c < cpr; c += WAVE) dst[c] = src[c];
            } else {
                // class-token row: same fused residual and LayerNorm as the merged rows (cpr <= 128 here)
                const int c0 = lane, c1 = WAVE + lane;
                const bool a0 = c0 < cpr, a1 = c1 < cpr;
                Pack<TX, VEC> p0, p1;
                uint4 r0 = a0 ? src[c0] : uint4{0, 0, 0, 0}, r1 = a1 ? src[c1] : uint4{0, 0, 0, 0};
                __builtin_memcpy(&p0, &r0, 16);
                __builtin_memcpy(&p1, &r1, 16);
                if (ln.a_own ? ln.cls_addend != nullptr : ln.addend != nullptr) {
                    const uint4 *asrc = ln.a_own
                        ? reinterpret_cast<const uint4 *>(reinterpret_cast<const TX *>(ln.cls_addend) + b * (int64_t)C)
                        : reinterpret_cast<const uint4 *>(reinterpret_cast<const TX *>(ln.addend) + b * lin.outer_stride);
                    Pack<TX, VEC> q0, q1;
                    uint4 s0 = a0 ? asrc[c0] : uint4{0, 0, 0, 0}, s1 = a1 ? asrc[c1] : uint4{0, 0, 0, 0};
                    __builtin_memcpy(&q0, &s0, 16);
                    __builtin_memcpy(&q1, &s1, 16);
                    p0 = add_packs<TX, VEC>(p0, q0);
                    p1 = add_packs<TX, VEC>(p1, q1);
                    __builtin_memcpy(&r0, &p0, 16);
                    __builtin_memcpy(&r1, &p1, 16);
                }
                if (ln.xbias) {
                    const TX *xb = reinterpret_cast<const TX *>(ln.xbias);
                    if (a0) dst[c0] = add_xbias<TX, VEC>(r0, xb + c0 * VEC);
                    if (a1) dst[c1] = add_xbias<TX, VEC>(r1, xb + c1 * VEC);
                } else {
                    if (a0) dst[c0] = r0;
                    if (a1) dst[c1] = r1;
                }
                float t = 0.0f, u = 0.0f;
#pragma unroll
                for (int e = 0; e < VEC; ++e) t += (a0 ? to_f32(p0.e[e]) : 0.0f) + (a1 ? to_f32(p1.e[e]) : 0.0f);
                const float fc = (float)C;
                const float m = wave_sum(t) / fc;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {  // centred second pass over the registers, like every other row
                    const float d0 = a0 ? to_f32(p0.e[e]) - m : 0.0f, d1 = a1 ? to_f32(p1.e[e]) - m : 0.0f;
                    u = __fmaf_rn(d0, d0, u);
                    u = __fmaf_rn(d1, d1, u);
                }
                const float rs = 1.0f / __builtin_sqrtf(wave_sum(u) / fc + ln.eps);
                const TX *lw = reinterpret_cast<const TX *>(ln.weight), *lb = reinterpret_cast<const TX *>(ln.bias);
                uint4 *ydst = reinterpret_cast<uint4 *>(reinterpret_cast<TX *>(ln.y) + b * lout.outer_stride);
                if (a0) {
                    float w8[VEC], b8[VEC];
                    load_pack<TX, VEC>(lw + c0 * VEC, w8);
                    load_pack<TX, VEC>(lb + c0 * VEC, b8);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) p0.e[e] = from_f32<TX>((to_f32(p0.e[e]) - m) * rs * w8[e] + b8[e]);
                    __builtin_memcpy(&r0, &p0, 16);
                    ydst[c0] = r0;
                }
                if (a1) {
                    float w8[VEC], b8[VEC];
                    load_pack<TX, VEC>(lw + c1 * VEC, w8);
                    load_pack<TX, VEC>(lb + c1 * VEC, b8);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) p1.e[e] = from_f32<TX>((to_f32(p1.e[e]) - m) * rs * w8[e] + b8[e]);
                    __builtin_memcpy(&r1, &p1, 16);
                    ydst[c1] = r1;
                }
            }
        }
        return;
    }
    const int g = gy;
    if (lw >= rg_per_group) {
        // edge waves: one per (group, rank k).  The wave of the FIRST edge into a destination builds that
        // row (all its sources, rank order); the others leave.  Destinations with sources therefore never
        // hold up the streaming waves above, and run concurrently with them.
        const int k = lw - rg_per_group;
        if (OP == OP_DROP || k >= r) return;
        const int64_t *dstg = dst_idx + (int64_t)g * r;
        if (r <= WAVE && cpr <= 2 * WAVE) {
            // the whole index list in one lane each: two memory round trips instead of five (merge_dst_row_r64)
            const int T1e = (T_ + 1) >> 1;
            const bool add = LN && ln.addend;
            merge_dst_row_r64<TX, TS, VEC, OP, LN>(
                group_ptr(x, lin, g), size ? size + (int64_t)g * T_ : nullptr, C, lin.tok_stride, r, g, k,
                src_idx + (int64_t)g * r, dstg, keep, group_ptr(xout, lout, g), lout.tok_stride,
                sout ? sout + (int64_t)g * To : nullptr, lsout ? lsout + (int64_t)g * To : nullptr, T1e - r, distill, lane,
                &ln, LN ? group_ptr(reinterpret_cast<TX *>(ln.y), lout, g) : nullptr,
                add ? (ln.a_own ? group_ptr(reinterpret_cast<const TX *>(ln.addend), ln.la, g)
                                : group_ptr(reinterpret_cast<const TX *>(ln.addend), lin, g)) : nullptr,
                ln.a_own ? ln.la.tok_stride : lin.tok_stride);
            return;
        }
        const int j = (int)dstg[k];
        bool earlier = false;
        for (int base = 0; base < k; base += WAVE) {
            const int kk = base + lane;
            earlier = earlier || (__ballot((kk < k) && ((int)dstg[kk] == j)) != 0ull);
        }
        if (earlier) return;
        const int T1e = (T_ + 1) >> 1, Ue = T1e - r;
        const int o = out_row_dst(j, Ue, distill);
        merge_dst_row<TX, TS, VEC, OP, LN>(
            group_ptr(x, lin, g), size ? size + (int64_t)g * T_ : nullptr, C, lin.tok_stride, r, g, j,
            src_idx + (int64_t)g * r, dstg, keep, group_ptr(xout, lout, g) + (int64_t)o * lout.tok_stride,
            sout ? sout + (int64_t)g * To + o : nullptr, lane, &ln,
            LN ? group_ptr(reinterpret_cast<TX *>(ln.y), lout, g) + (int64_t)o * lout.tok_stride : nullptr,
            (LN && ln.addend) ? (ln.a_own ? group_ptr(reinterpret_cast<const TX *>(ln.addend), ln.la, g)
                                          : group_ptr(reinterpret_cast<const TX *>(ln.addend), lin, g)) : nullptr,
            lsout ? lsout + (int64_t)g * To + o : nullptr, ln.a_own ? ln.la.tok_stride : lin.tok_stride);
        return;
    }
    const int o0 = lw * R;
    const int T1 = (T_ + 1) >> 1, U = T1 - r;
    const TX *xg = group_ptr(x, lin, g);
    TX *og = group_ptr(xout, lout, g);
    const TS *sg = size ? size + (int64_t)g * T_ : nullptr;
    const int64_t *dstg = dst_idx ? dst_idx + (int64_t)g * r : nullptr;

    // per-row facts, computed by lanes 0..R-1 in parallel and broadcast as wave-uniform scalars: source
    // token, B-row number, "something merges into it", size.  Load order matters for latency: the first block
    // of dst_idx and the unm_idx entries go out together; the sizes are requested as soon as the tokens are
    // known but only READ after the token rows' own loads have been issued, so a wave waits for two memory
    // round trips (index -> rows), not three.  (Several R-row slabs per wave behind ONE index round trip were
    // measured in round 2: no gain for this kernel, a loss for the plain merge -- the kernel is bound by its vector
    // instructions, not by latency.)
    const int d_first = (OP != OP_DROP && lane < r) ? (int)dstg[lane] : -2;
    // (branch-free: every lane computes, lanes that own no row read a harmless entry -- no divergent control flow
    // and no wait between the index loads and the row loads)
    float my_s = 1.0f;
    const bool my_valid = lane < R && (o0 + lane) < To;
    bool is_dst;
    int idx;
    decode_out_row(my_valid ? o0 + lane : 0, U, distill, is_dst, idx);
    int my_tok = 2 * idx + 1, my_j = idx;
    // a wave whose rows are all destination (odd) tokens knows its tokens without reading an index: its row loads
    // depend on no earlier load at all
    const bool all_dst = (!distill && o0 >= U) || o0 > U;
    if (U > 0 && !all_dst) {  // (wave-uniform; unm_idx may be absent when every even token is merged away)
        const int ut = 2 * (int)unm_idx[(int64_t)g * U + ((my_valid && !is_dst) ? idx : 0)];
        my_tok = is_dst ? my_tok : ut;
        my_j = is_dst ? my_j : -1;
    }
    my_tok = my_valid ? my_tok : 0;
    my_j = my_valid ? my_j : -1;
    // (the size is read unconditionally -- from the tokens themselves when there are no sizes -- so that no branch
    // invites the compiler to wait for it before the row loads have been issued)
    const bool load_size = (OP == OP_WAVG) && sg;  // wave-uniform
    const TS my_s_raw = (load_size ? sg : reinterpret_cast<const TS *>(xg))[my_tok];
    const unsigned long long vmask = __ballot(my_valid);
    const int tok0 = __builtin_amdgcn_readlane(my_tok, 0), tok1 = __builtin_amdgcn_readlane(my_tok, 1),
              tok2 = __builtin_amdgcn_readlane(my_tok, 2), tok3 = __builtin_amdgcn_readlane(my_tok, 3);
    const int j0 = __builtin_amdgcn_readlane(my_j, 0), j1 = __builtin_amdgcn_readlane(my_j, 1),
              j2 = __builtin_amdgcn_readlane(my_j, 2), j3 = __builtin_amdgcn_readlane(my_j, 3);
    // rows that exist; which of them receive sources (and are left to the edge waves) is decided AFTER the row loads
    // have been issued -- a row that turns out to be one is read for nothing (r of T rows), but no row load waits for
    // dst_idx
    bool va0 = vmask & 1ull, va1 = vmask & 2ull, va2 = vmask & 4ull, va3 = vmask & 8ull;
    if (EAGER && OP != OP_DROP) {  // (r <= 64: the host's condition -- every dst_idx entry of the group is in d_first)
        va0 = va0 && (__ballot(d_first == j0) == 0ull);
        va1 = va1 && (__ballot(d_first == j1) == 0ull);
        va2 = va2 && (__ballot(d_first == j2) == 0ull);
        va3 = va3 && (__ballot(d_first == j3) == 0ull);
    }

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
        const bool ok = (rr == 0 ? va0 : (rr == 1 ? va1 : (rr == 2 ? va2 : va3))) && (q < total);
        rowof[it] = ok ? rr : -1;
        // unconditional: a lane without a chunk re-reads the start of the wave's first row (always a real token) and
        // ignores it -- no control flow around the loads, so all of them are in flight together
        raw[it] = ld16(reinterpret_cast<const char *>(xg + (int64_t)(ok ? t : tok0) * lin.tok_stride) + (ok ? cc : 0) * 16);
    }
    uint4 rawa[NIT];
    if (LN && ln.addend) {  // fused residual: the rows that are merged are round(x + addend)
        const TX *agp = ln.a_own ? group_ptr(reinterpret_cast<const TX *>(ln.addend), ln.la, g)
                                 : group_ptr(reinterpret_cast<const TX *>(ln.addend), lin, g);
        const int64_t astride = ln.a_own ? ln.la.tok_stride : lin.tok_stride;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rr = rowof[it];
            const int q = it * WAVE + lane;
            const int cc = rr < 0 ? 0 : q - rr * cpr;
            const int t = rr <= 0 ? tok0 : (rr == 1 ? tok1 : (rr == 2 ? tok2 : tok3));
            rawa[it] = ld16(reinterpret_cast<const char *>(agp + (int64_t)t * astride) + cc * 16);
        }
    }
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
    const bool ok0 = va0 && !e0, ok1 = va1 && !e1, ok2 = va2 && !e2, ok3 = va3 && !e3;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rr = rowof[it];
        const bool ok = rr >= 0 && (rr == 0 ? ok0 : (rr == 1 ? ok1 : (rr == 2 ? ok2 : ok3)));
        rowof[it] = ok ? rr : -1;
    }
    if (LN && ln.addend) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (rowof[it] < 0) continue;
            Pack<TX, VEC> px, pa;
            __builtin_memcpy(&px, &raw[it], 16);
            __builtin_memcpy(&pa, &rawa[it], 16);
            const Pack<TX, VEC> ps = add_packs<TX, VEC>(px, pa);
            __builtin_memcpy(&raw[it], &ps, 16);
        }
    }
    // the bias folded into x' (xbias): its chunks are requested here, ahead of the stores that need them
    uint4 xbraw[NIT];
    if (LN && ln.xbias) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rr = rowof[it];
            xbraw[it] = *(reinterpret_cast<const uint4 *>(ln.xbias) + (rr < 0 ? 0 : it * WAVE + lane - rr * cpr));
        }
    }
    my_s = (load_size && my_valid) ? to_f32(my_s_raw) : 1.0f;
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
        char *xdst = reinterpret_cast<char *>(og + (int64_t)(o0 + rr) * lout.tok_stride) + cc * 16;
        if (LN && ln.xbias)
            st16(xdst, add_xbias<TX, VEC>(outv, reinterpret_cast<const TX *>(&xbraw[it])));
        else
            st16(xdst, outv);
        if (LN) raw[it] = outv;  // keep the merged bits: LayerNorm runs on x' itself (what is written without xbias)
    }
    if constexpr (LN) {
        // LayerNorm of the (up to) four rows of this wave as they were stored (ln_rows above)
        TX *yg = group_ptr(reinterpret_cast<TX *>(ln.y), lout, g);
        const int64_t ystride = lout.tok_stride;
        ln_rows<TX, NIT>(raw, rowof, R, cpr, C, ln.eps, reinterpret_cast<const TX *>(ln.weight),
                         reinterpret_cast<const TX *>(ln.bias), lane,
                         [&](int, int rr, int cc, const uint4 &yv) __attribute__((always_inline)) {
                             st16(reinterpret_cast<char *>(yg + (int64_t)(o0 + rr) * ystride) + cc * 16, yv);
                         });
    }
    if (OP == OP_WAVG && lane < R && my_valid) {
        const bool mine_has_edges = lane == 0 ? e0 : (lane == 1 ? e1 : (lane == 2 ? e2 : e3));
        if (!mine_has_edges)
            store_size<TS>(sout + (int64_t)g * To + o0 + lane, lsout ? lsout + (int64_t)g * To + o0 + lane : nullptr, my_s);
    }
}

// k_add_ln_rows: the other residual of the patched block, fused with the LayerNorm that reads its result:
//     x = x + mlp(norm2(x))            (tome/patch/videomae.py:29, end of ToMeBlock.forward)
//     ... next block: self.norm1(x)    (tome/patch/videomae.py:19)
// x' = round(x + a) and y = LayerNorm(x') in one pass (same arithmetic as the LN tail of k_merge_rows_fast).
// A wave owns R consecutive rows, NIT 16-byte chunks per lane, everything loaded before use.
template <typename TX, int NIT>
__global__ __launch_bounds__(256) void k_add_ln_rows(const TX *__restrict__ x, const TX *__restrict__ a,
                                                     int64_t rows, int C, int R, int cpr, LnArgs ln,
                                                     TX *__restrict__ xout) {
    constexpr int VEC = 16 / sizeof(TX);
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row0 = w * R;
    if (row0 >= rows) return;
    const int nrow = (int)((rows - row0) < R ? (rows - row0) : R);
    const int total = nrow * cpr;
    // a == nullptr: LayerNorm only (x is the finished sum -- the caller's GEMM accumulated onto it), nothing is
    // written to xout
    const bool add = a != nullptr;
    const uint4 *xs = reinterpret_cast<const uint4 *>(x + row0 * C),
                *as = reinterpret_cast<const uint4 *>((add ? a : x) + row0 * C);
    uint4 raw[NIT], rawa[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = it * WAVE + lane;
        if (q < total) {
            raw[it] = ld16(xs + q);
            if (add) rawa[it] = ld16(as + q);
        }
    }
    uint4 *xo = reinterpret_cast<uint4 *>(xout + row0 * C);
    int rowof[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = it * WAVE + lane;
        const int rr = (q >= cpr) + (q >= 2 * cpr) + (q >= 3 * cpr);
        rowof[it] = q < total ? rr : -1;
        if (q >= total || !add) continue;
        Pack<TX, VEC> ps, pa;
        __builtin_memcpy(&ps, &raw[it], 16);
        __builtin_memcpy(&pa, &rawa[it], 16);
        ps = add_packs<TX, VEC>(ps, pa);
        __builtin_memcpy(&raw[it], &ps, 16);
        st16(xo + q, raw[it]);
    }
    // y_group > 0: rows come in groups of y_group whose FIRST row (a class token) has no place in y -- y holds the other
    // y_group - 1 rows of every group, compacted (TimeSformer's temporal_norm1 feeds only the patch tokens,
    // tome/patch/timesformer.py:24-26, so `xn[:, 1:]` regrouped '(b p) t m' is a view instead of a copy)
    const int yg = ln.y_group;
    TX *yb = reinterpret_cast<TX *>(ln.y);
    ln_rows<TX, NIT>(raw, rowof, nrow, cpr, C, ln.eps, reinterpret_cast<const TX *>(ln.weight),
                     reinterpret_cast<const TX *>(ln.bias), lane,
                     [&](int, int rr, int cc, const uint4 &yv) __attribute__((always_inline)) {
                         int64_t yrow = row0 + rr;
                         if (yg > 0) {
                             const int64_t gb = yrow / yg;
                             if (yrow - gb * yg == 0) return;
                             yrow -= gb + 1;
                         }
                         st16(reinterpret_cast<uint4 *>(yb + yrow * C) + cc, yv);
                     });
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
// Source tracking, first layer (merge.py:372-384 with source=None): the reference merges an identity matrix
// with mode "max" -- output row o, column t is 1 exactly when token t ends up in merged row o.  That is the
// row map of the matching, so the [n,T,T] identity and the amax over its zeros are never made:
//   k_row_map      src/dst/unm_idx -> row_map[n,T1] (for plans that were matched without one)
//   k_source_init  out[g,o,t] = (row of token t == o), one coalesced pass over the [n,T-r,T] result;
//                  drop mode (merge.py:215-271 applied to the identity): merged-away tokens have no row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_map(int n, int T1, int r, int distill,
                                                 const int64_t *__restrict__ src_idx,
                                                 const int64_t *__restrict__ dst_idx,
                                                 const int64_t *__restrict__ unm_idx, int *__restrict__ row_map) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * T1) return;
    const int g = (int)(i / T1), k = (int)(i - (int64_t)g * T1), U = T1 - r;
    if (k < r) row_map[(int64_t)g * T1 + (int)src_idx[(int64_t)g * r + k]] = out_row_dst((int)dst_idx[(int64_t)g * r + k], U, distill);
    else row_map[(int64_t)g * T1 + (int)unm_idx[(int64_t)g * U + (k - r)]] = out_row_unm(k - r, distill);
}

__global__ __launch_bounds__(256) void k_source_init(int n, int T_, int r, int distill, int drop,
                                                     const int *__restrict__ row_map, float *__restrict__ out) {
    const int To = T_ - r, T1 = (T_ + 1) >> 1, U = T1 - r;
    const int64_t row = blockIdx.x;  // (g, o)
    const int g = (int)(row / To), o = (int)(row - (int64_t)g * To);
    const int *rm = row_map + (int64_t)g * T1;
    float *orow = out + row * T_;
    for (int t = threadIdx.x; t < T_; t += blockDim.x) {
        int rt;
        bool live = true;
        if (t & 1) rt = out_row_dst(t >> 1, U, distill);
        else {
            rt = rm[t >> 1];
            if (drop) {
                bool is_dst;
                int idx;
                decode_out_row(rt, U, distill, is_dst, idx);
                live = !is_dst;  // an even token that maps onto a destination row was merged away: dropped
            }
        }
        orow[t] = (live && rt == o) ? 1.0f : 0.0f;
    }
}

// k_add_ln_regroup: the middle of TimeSformer's divided space-time block (tome/patch/timesformer.py:24-38):
//     xt = x[:, 1:] + temporal_fc(res_temporal)                       residual of the temporal attention
//     xs = cat(cls replicated per frame, 'b (p t) m -> (b t) p m' of xt) spatial regrouping
//     ... self.norm1(xs)
// One pass: x1 = cat(cls, xt) in the token layout [B, 1 + P*F, C], and y = norm1 of every row written straight to
// its place in the regrouped [B*F, 1 + P, C] tensor (a LayerNorm is per row, so normalising before the permutation
// is the same thing); the class row is normalised once and stored F times.  Replaces an add, a strided copy and a
// LayerNorm (7 passes over the tokens) by 2 reads + 2 writes.  Same arithmetic as k_add_ln_rows.
template <typename TX, int NIT>
__global__ __launch_bounds__(256) void k_add_ln_regroup(const TX *__restrict__ x, const TX *__restrict__ a, int B, int F,
                                                        int P, int C, int R, int cpr, LnArgs ln,
                                                        TX *__restrict__ xout) {
    constexpr int VEC = 16 / sizeof(TX);
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = 1 + P * F;
    const int64_t rows = (int64_t)B * N;
    const int64_t row0 = w * R;
    if (row0 >= rows) return;
    const int nrow = (int)((rows - row0) < R ? (rows - row0) : R);
    const int total = nrow * cpr;
    // per-row facts (wave-uniform): addend row (-1: class token, nothing is added) and the row of y
    int64_t arow[FAST_MAXR], yrow[FAST_MAXR];
#pragma unroll
    for (int rr = 0; rr < FAST_MAXR; ++rr) {
        const int64_t gr = row0 + (rr < nrow ? rr : 0);
        const int64_t b = gr / N;
        const int k = (int)(gr - b * N);
        if (k == 0) {
            arow[rr] = -1;
            yrow[rr] = b * F * (1 + P);  // frame 0; frames 1..F-1 follow at (1+P) rows each
        } else {
            const int p = (k - 1) / F, t = (k - 1) - p * F;
            arow[rr] = b * (int64_t)(P * F) + (k - 1);
            yrow[rr] = (b * F + t) * (int64_t)(1 + P) + 1 + p;
        }
    }
    const uint4 *xs = reinterpret_cast<const uint4 *>(x + row0 * C);
    uint4 raw[NIT], rawa[NIT];
    int rowof[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = it * WAVE + lane;
        const int rr = (q >= cpr) + (q >= 2 * cpr) + (q >= 3 * cpr);
        rowof[it] = q < total ? rr : -1;
        if (q >= total) continue;
        raw[it] = ld16(xs + q);
        const int64_t ar = rr == 0 ? arow[0] : (rr == 1 ? arow[1] : (rr == 2 ? arow[2] : arow[3]));
        if (ar >= 0) rawa[it] = ld16(reinterpret_cast<const uint4 *>(a + ar * C) + (q - rr * cpr));
    }
    uint4 *xo = reinterpret_cast<uint4 *>(xout + row0 * C);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rr = rowof[it];
        if (rr < 0) continue;
        const int q = it * WAVE + lane;
        const int64_t ar = rr == 0 ? arow[0] : (rr == 1 ? arow[1] : (rr == 2 ? arow[2] : arow[3]));
        if (ar >= 0) {
            Pack<TX, VEC> ps, pa;
            __builtin_memcpy(&ps, &raw[it], 16);
            __builtin_memcpy(&pa, &rawa[it], 16);
            ps = add_packs<TX, VEC>(ps, pa);
            __builtin_memcpy(&raw[it], &ps, 16);
        }
        st16(xo + q, raw[it]);
    }
    TX *yb = reinterpret_cast<TX *>(ln.y);
    ln_rows<TX, NIT>(raw, rowof, nrow, cpr, C, ln.eps, reinterpret_cast<const TX *>(ln.weight),
                     reinterpret_cast<const TX *>(ln.bias), lane,
                     [&](int, int rr, int cc, const uint4 &yv) __attribute__((always_inline)) {
                         const int64_t ar = rr == 0 ? arow[0] : (rr == 1 ? arow[1] : (rr == 2 ? arow[2] : arow[3]));
                         const int64_t yr = rr == 0 ? yrow[0] : (rr == 1 ? yrow[1] : (rr == 2 ? yrow[2] : yrow[3]));
                         uint4 *yp = reinterpret_cast<uint4 *>(yb + yr * C) + cc;
                         st16(yp, yv);
                         if (ar < 0)  // class token: the same normalised row in front of every frame's tokens
                             for (int t = 1; t < F; ++t) st16(yp + (int64_t)t * (1 + P) * cpr, yv);
                     });
}

// k_gelu_erf: the activation between the two GEMMs of the MLP the patched block calls (`self.mlp(self.norm2(x))`,
// tome/patch/videomae.py:29; nn.GELU() = exact erf form in VideoMAE / TimeSformer / Motionformer):
//     y = x * 0.5 * (1 + erf(x / sqrt(2)))      fp32 arithmetic on 16-bit values, one rounding -- the expression and
// operation order of the framework's kernel, so the result is bit-identical to it.  A pure streaming pass: four
// 16-byte chunks per lane in flight, non-temporal both ways (the [tokens, 4C] activation is 1.2 GB at batch 128).
template <typename TX>
__global__ __launch_bounds__(256) void k_gelu_erf(const TX *__restrict__ x, TX *__restrict__ y, int64_t chunks) {
    constexpr int VEC = 16 / sizeof(TX);
    constexpr int NIT = 4;
    const int64_t base = ((int64_t)blockIdx.x * blockDim.x) * NIT + threadIdx.x;
    uint4 raw[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int64_t q = base + (int64_t)it * blockDim.x;
        if (q < chunks) raw[it] = ld16(reinterpret_cast<const uint4 *>(x) + q);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int64_t q = base + (int64_t)it * blockDim.x;
        if (q >= chunks) continue;
        Pack<TX, VEC> pk;
        __builtin_memcpy(&pk, &raw[it], 16);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float v = to_f32(pk.e[e]);
            pk.e[e] = from_f32<TX>(v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)));
        }
        uint4 o;
        __builtin_memcpy(&o, &pk, 16);
        st16(reinterpret_cast<uint4 *>(y) + q, o);
    }
}
