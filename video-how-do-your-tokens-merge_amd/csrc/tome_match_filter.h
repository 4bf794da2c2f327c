// tome_match_filter.h -- part of csrc/tome_kernels.hip: the similarity + row-max stage of the matching for bf16 keys
// WITHOUT evaluating every score on the fp32 matrix pipe.
//
// Contract (unchanged, oracle/tome_oracle.c): S[i][j] = the fp32 fma chain over k of u_i[k] * u_j[k] with
// u = fdiv(v, ||v||) (v = the bf16 metric / head mean as fp32), node_max[i] = max_j S[i][j], node_idx[i] = the first j
// that attains it.  k_scores_rowmax evaluates all T1 x T2 scores with v_mfma_f32_32x32x2_f32 (exact fp32 chain, 1/16
// of the bf16 matrix rate) and sits at 0.67 of that pipe's peak.  Only the WINNER of every row has to be exact:
//
//   1. k_scores_filter: approximate scores S~ = (v_i . v_j) * (1/||v_j||) on v_mfma_f32_32x32x16_bf16 -- the operands
//      ARE bf16, so the products are exact and only the fp32 accumulation order and the missing division differ from
//      the contract: |S~ * (1/||v_i||) - S| <= EPS (bound below).  The row scale 1/||v_i|| does not move a row's
//      argmax, so it is applied to the window instead.  One wave per (group, 32-row A tile) sweeps all B tiles once
//      and records, per lane, every column whose S~ lies within the window of the running maximum -- a superset of
//      the columns within the window of the FINAL maximum (the running maximum only grows); when the maximum moves up
//      by more than the window the list starts over (everything recorded before is out of reach for good).
//   2. k_exact_rows: per row, the recorded columns that lie within the window of the final approximate maximum
//      (normally one, two for near-ties) are evaluated with the contract's own arithmetic (fdiv + fma chain in k
//      order); the largest exact score wins, the smallest column among equals -- the same bits k_scores_rowmax gives.
//   3. A lane whose list overflows (tie-heavy or monotone data: more than FILT_KH near-records in its half row) flags
//      its tile, and so does a norm outside [FILT_NORM_LO, FILT_NORM_HI]; a flagged tile is recomputed on the fp32
//      matrix pipe with k_scores_rowmax's arithmetic by a wave of the k_exact_rows launch (fp32_pass_of_tile) -- the
//      filter degrades to the full fp32 pass, never to a different answer.
//   Launches of one matching on this path: k_unit_rows_f, k_scores_filter, k_exact_rows (+ its fallback waves),
//   k_rank_select -- four, none of them idle, no flag word to clear (round 4; rounds 2-3 ran seven: a memset, two
//   fallback kernels that returned at once, and k_scores_rowmax on the flagged tiles).
//
// Error bound.  v are bf16 (8 significant bits): every product v_i[k] v_j[k] is exact in fp32.  With D <= 64 terms the
// bf16 MFMA's fp32 accumulation differs from the exact dot product by <= 64 * 2^-24 * sum|v_i[k] v_j[k]| <= 3.9e-6
// * ||v_i|| ||v_j||; the contract's chain (divisions rounded to fp32, 64 fma) differs from the exact cosine by
// <= (2 * 2^-24 + 64 * 2^-24) <= 4.0e-6; the reciprocal 1/||v_j|| (v_rcp_f32, 1 ulp) and the scaling add <= 2 * 2^-23.
// In units of the cosine: |S~/||v_i|| - S| <= 8.2e-6.  FILT_WINDOW = 4e-5 on the cosine scale (>= 2 * EPS with a factor
// 2.4 in hand) decides which columns are kept; it is applied as FILT_WINDOW * ||v_i|| to the un-normalised S~.
#pragma once

#define FILT_KH 4            // records per lane (= per half row): (B tile, 16-bit mask of its columns) pairs
#define FILT_WINDOW 4e-5f
// Norms the bound above holds for.  Below LO the products v_i[k] v_j[k] of two such tokens reach the denormal range
// (flushed inside the matrix instruction: 64 flushed terms of < 2^-126 against ||v_i|| ||v_j|| >= 1e-28 stay below 1e-8
// of the cosine; with 1e-30, round 3's bound, they did not); above HI a dot product of two such tokens may overflow.
#define FILT_NORM_LO 1e-14f
#define FILT_NORM_HI 1e18f

typedef __bf16 filt_bf16x8 __attribute__((ext_vector_type(8)));

// A record: B tile jt in the upper half, in the lower half one bit per accumulator register v of the lane -- column
// j = 32 jt + (v & 3) + 8 (v >> 2) + 4 h (bit 15 - v: the bits are shifted in in register order).
typedef unsigned int CandEntry;

// 16-byte chunk index of (tile, ks, lane) in a bf16 fragment plane: tile = 32 rows x 64 channels = 4 KB,
// lane (row & 31) + 32 * hf holds channels 16 * ks + 8 * hf .. + 7 of its row
__device__ __forceinline__ int64_t bfrag_index(int64_t tile, int ks, int lane) { return (tile * 4 + ks) * 64 + lane; }

// ------------------------------------------------------------------------------------------------
// k_unit_rows_heads_f: k_unit_rows_heads for the filter path -- same head mean (fp32 sum in head order, / H, one
// rounding to bf16) and the same squared norm and flags, but instead of the fp32 unit vectors it leaves the bf16 mean
// itself in MFMA fragment order, the norm and its reciprocal: 40 MB instead of 145 MB per launch at batch 384.
// HEADS = false: the metric is given (tome_match, D <= 64, D % 8 == 0): no head loop.
// ------------------------------------------------------------------------------------------------
template <bool HEADS>
__global__ __launch_bounds__(256) void k_unit_rows_f(const bf16_t *__restrict__ keys, int64_t stride_n, int inner,
                                                     int64_t stride_inner, int64_t stride_h, int64_t stride_t, int n,
                                                     int H, int T_, int D, uint4 *__restrict__ vA,
                                                     uint4 *__restrict__ vB, int ntA, int ntB,
                                                     float *__restrict__ normA, float *__restrict__ normB,
                                                     float *__restrict__ invB, int T2p, uint8_t *__restrict__ badA,
                                                     uint8_t *__restrict__ badB) {
    const int lane = threadIdx.x & 63;
    const int b8 = lane & 7;
    const int64_t tok = ((int64_t)blockIdx.x * (blockDim.x >> 3)) + (threadIdx.x >> 3);
    const int64_t ntok = (int64_t)n * T_;
    const bool live = tok < ntok;
    const int64_t tk = live ? tok : ntok - 1;
    const int g = (int)((uint32_t)tk / (uint32_t)T_);
    const int t = (int)((uint32_t)tk - (uint32_t)g * (uint32_t)T_);
    const bf16_t *row = keys + (int64_t)t * stride_t + 8 * b8;
    if (inner == 1) row += (int64_t)g * stride_n;
    else row += (int64_t)(g / inner) * stride_n + (int64_t)(g % inner) * stride_inner;
    float v[8];
    const bool in_range = 8 * b8 < D;  // (D < 64: the padding channels are zeros)
    if (HEADS) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll 12
        for (int h = 0; h < H; ++h) {
            float kv[8];
            Load8<bf16_t>::run(row + (int64_t)h * stride_h, kv);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = __fadd_rn(acc[e], kv[e]);
        }
        const float fh = (float)H;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = to_f32(from_f32<bf16_t>(__fdiv_rn(acc[e], fh)));  // k.mean(1) in bf16
    } else {
        if (in_range) Load8<bf16_t>::run(row, v);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.0f;
        }
    }
    float part = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) part = __fmaf_rn(v[e], v[e], part);
    float ss = 0.0f;
    const int base = lane & ~7;
    const int nblk = D >> 3;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const float p = __shfl(part, base + l);
        if (l < nblk) ss = __fadd_rn(ss, p);  // blocks in ascending order, as k_unit_rows
    }
    const float nr = __builtin_sqrtf(ss);
    // would the unit vector hold a NaN?  (merge.py:51 has no epsilon: a zero / inf / NaN token.)  The contract's
    // division v / ||v|| is a NaN exactly when an operand is one, or both are zero, or both are infinite (IEEE 754) --
    // decided from the operands: the eight correctly rounded divisions per lane this replaced were a fifth of the
    // kernel's vector instructions, and only their NaN-ness was used
    bool nan_here = false;
    if (in_range) {
        const bool n_nan = nr != nr, n_zero = nr == 0.0f, n_inf = __builtin_isinf(nr);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            nan_here = nan_here || (v[e] != v[e]) || n_nan || (n_zero && v[e] == 0.0f) || (n_inf && __builtin_isinf(v[e]));
    }
    const int rowi = t >> 1;
    const bool odd = t & 1;
    if (live) {
        Pack<bf16_t, 8> pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) pk.e[e] = from_f32<bf16_t>(v[e]);  // exact: v is a bf16 value
        uint4 raw;
        __builtin_memcpy(&raw, &pk, 16);
        uint4 *dst = odd ? vB + (int64_t)g * ntB * 256 : vA + (int64_t)g * ntA * 256;
        dst[bfrag_index(rowi >> 5, b8 >> 1, (rowi & 31) + 32 * (b8 & 1))] = raw;
    }
    const unsigned long long nan_mask = __ballot(nan_here);
    if (live && b8 == 0) {
        const uint8_t flag = ((nan_mask >> (lane & ~7)) & 0xFFull) ? 1 : 0;
        // (a usable token whose norm is so small or large that 1/||v|| or its products may leave the normal fp32 range:
        // k_scores_filter reads the norms it works with and sends the tiles concerned to the fp32 pass itself --
        // FILT_NORM_LO / FILT_NORM_HI there; nothing is flagged here, so no flag word needs clearing before this launch)
        if (odd) {
            badB[(int64_t)g * (T_ >> 1) + rowi] = flag;
            normB[(int64_t)g * (T_ >> 1) + rowi] = nr;
            // (a NaN / zero / inf token: its scores are NaN in the contract and must stay out of the running maximum)
            invB[(int64_t)g * T2p + rowi] = flag ? __builtin_nanf("") : __builtin_amdgcn_rcpf(nr);
        } else {
            badA[(int64_t)g * ((T_ + 1) >> 1) + rowi] = flag;
            normA[(int64_t)g * ((T_ + 1) >> 1) + rowi] = nr;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_scores_filter: see the top of this file.  One single-wave workgroup per (group, FILT_ATW consecutive A tiles):
// every B tile a wave fetches from L2 (4 KB) meets FILT_ATW x 32 rows -- with one A tile per wave the kernel moved 2 GB
// per launch through L2 (B tiles + the reciprocal norms, re-read by each of a group's 25 A tiles) and ran at L2 speed
// (136 us at batch 384), four times what its matrix instructions need.  The group's reciprocal norms sit in LDS (one
// coalesced read per wave), the fragments' 16 per lane come back as four broadcast ds_read_b128 per tile.
// XCD-aware numbering as k_scores_rowmax (the waves of a group stream the same B tiles: ids congruent mod 8).
// Accumulator register v of lane l: column j = 32 jt + (v & 3) + 8 (v >> 2) + 4 (l >> 5) of row i = 32 ti + (l & 31).
// ------------------------------------------------------------------------------------------------
#ifndef FILT_ATW
#define FILT_ATW 5  // (measured at batch 384 x 1568 tokens, 25 A tiles: 2 -> 72 us, 3 -> 80, 4 -> 77, 5 -> 65: five equal waves per group)
#endif
#define FILT_MAX_T2P 8192  // reciprocal norms of one group in LDS (32 KB); longer sequences keep the fp32 pass

__global__ __launch_bounds__(64) void k_scores_filter(const uint4 *__restrict__ vA, const uint4 *__restrict__ vB,
                                                      const float *__restrict__ normA, const float *__restrict__ invB,
                                                      int n, int T1, int T2, int T2p, int ntA, int ntB,
                                                      int distill_token, CandEntry *__restrict__ cand,
                                                      uint8_t *__restrict__ cand_n, uint8_t *__restrict__ tile_flag) {
    extern __shared__ __attribute__((aligned(16))) float s_inv[];  // [T2p]
    const int L = blockIdx.x;
    const int xcd = L & 7, qq = L >> 3;
    const int nwa = (ntA + FILT_ATW - 1) / FILT_ATW;  // waves per group
    const int g = (qq / nwa) * 8 + xcd;
    if (g >= n) return;
    const int ti0 = (qq % nwa) * FILT_ATW;
    const int lane = threadIdx.x;
    const int col = lane & 31, h = lane >> 5;

    // Norms the approximate scores cannot be trusted with: a usable token whose norm lies outside
    // [FILT_NORM_LO, FILT_NORM_HI] -- its reciprocal, or a product v_i[k] * v_j[k] / (||v_i|| ||v_j||) of two such tokens,
    // may leave the NORMAL fp32 range (denormals are flushed inside the bf16 matrix instruction), and then the error
    // bound of the header does not hold.  A column like that sends every tile of its group to the fp32 pass, a row like
    // that its own tile (never seen on real keys).  Decided here from the norms this wave reads anyway, per tile, with
    // plain stores: no flag word that would have to be cleared in front of the matching.
    bool b_out = false;
    for (int j = lane; j < T2p; j += 64) {
        const float iv = invB[(int64_t)g * T2p + j];
        s_inv[j] = iv;
        // (NaN = a token whose unit vector holds a NaN: handled by the NaN rule, not a range problem)
        b_out = b_out || (j < T2 && iv == iv && !(iv >= 1.0001f / FILT_NORM_HI && iv <= 0.9999f / FILT_NORM_LO));
    }
    const bool force_group = __ballot(b_out) != 0ull;
    __syncthreads();  // (one wave: orders the LDS writes in front of the other lanes' reads)

    filt_bf16x8 a[FILT_ATW][4];
    // Per lane and A tile: thr = (running maximum of S~) - window; cnt records in rec[0 .. ] (newest first).
    // Everything below is straight vector code: no compare feeds a branch or the execution mask inside the sweep
    // (the first form of this kernel tested every register's hits with a scalar branch: 17 vector -> scalar round
    // trips per tile, 1160 cycles per tile where the arithmetic needs 250).
    float window[FILT_ATW], thr[FILT_ATW];
    int cnt[FILT_ATW];
    unsigned rec[FILT_ATW][FILT_KH];
    bool row_ok[FILT_ATW];
    unsigned a_out = 0u;  // bit u: a row of A tile u has a norm outside the trusted range (wave-uniform: a scalar register)
#pragma unroll
    for (int u = 0; u < FILT_ATW; ++u) {
        const int ti = min(ti0 + u, ntA - 1);  // (a wave past the last tile repeats it and writes nothing)
        const uint4 *at = vA + ((int64_t)g * ntA + ti) * 256 + lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 raw = at[ks * 64];
            __builtin_memcpy(&a[u][ks], &raw, 16);
        }
        const int i = ti * TILE_ROWS + col;
        row_ok[u] = (ti0 + u < ntA) && i < T1;
        // the window on the scale of S~ = S * ||v_i||
        const float ni = row_ok[u] ? normA[(int64_t)g * T1 + i] : 1.0f;
        a_out |= (__ballot(!(ni >= FILT_NORM_LO && ni <= FILT_NORM_HI)) != 0ull) ? (1u << u) : 0u;
        window[u] = FILT_WINDOW * (row_ok[u] ? ni : 0.0f);
        thr[u] = -INFINITY;
        cnt[u] = 0;
#pragma unroll
        for (int c = 0; c < FILT_KH; ++c) rec[u][c] = 0u;
    }
    const uint4 *bt = vB + (int64_t)g * ntB * 256 + lane;
    uint4 braw[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) braw[ks] = bt[ks * 64];
    for (int jt = 0; jt < ntB; ++jt) {
        filt_bf16x8 b[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) __builtin_memcpy(&b[ks], &braw[ks], 16);
        const int nx = jt + 1 < ntB ? jt + 1 : jt;  // the next tile's operands (the last iteration re-reads its own)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) braw[ks] = bt[((int64_t)nx * 4 + ks) * 64];
        float iv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t4 = *reinterpret_cast<const f32x4 *>(s_inv + 32 * jt + 8 * q + 4 * h);
            iv[4 * q + 0] = t4.x; iv[4 * q + 1] = t4.y; iv[4 * q + 2] = t4.z; iv[4 * q + 3] = t4.w;
        }
        if ((jt == 0 && distill_token) || (jt + 1) * TILE_ROWS > T2) {  // wave-uniform: columns that do not exist
            const int jbase = jt * TILE_ROWS + 4 * h;                     // get a NaN scale -- never a hit, never a max
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int j = jbase + (v & 3) + 8 * (v >> 2);
                if (j >= T2 || (distill_token && j == 0)) iv[v] = __builtin_nanf("");
            }
        }
#pragma unroll
        for (int u = 0; u < FILT_ATW; ++u) {
            f32x16 acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks], a[u][ks], acc, 0, 0, 0);
            // the tile's scores and their maximum (a NaN score -- a bad or absent column -- is ignored by v_max) ...
            float sv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) sv[v] = acc[v] * iv[v];
            float tmax = fmaxf(fmaxf(sv[0], sv[1]), sv[2]);
#pragma unroll
            for (int v = 3; v < 15; v += 2) tmax = fmaxf(fmaxf(tmax, sv[v]), sv[v + 1]);
            tmax = fmaxf(tmax, sv[15]);
            const float th = thr[u];
            const float thn = fmaxf(th, tmax - window[u]);
            // ... then one bit per column above the threshold AFTER this tile: the sign bit of thn - S~ (strictly above;
            // the maximum itself gives -window < 0).  A NaN score leaves a NaN: whatever its sign bit says, the exact
            // pass drops absent and protected columns and columns whose exact score is NaN.
            unsigned mask = 0u;
#pragma unroll
            for (int v = 0; v < 16; ++v) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(thn - sv[v]), 31);
            // a maximum that has moved up by more than the window makes every recorded column stale (all of them are
            // <= the old maximum): the list starts over.  On ordinary data a lane's list therefore holds one record;
            // only near-ties -- or a maximum creeping up in steps smaller than the window -- accumulate.
            const bool jumped = thn > th + window[u];
            cnt[u] = jumped ? 0 : cnt[u];
            thr[u] = thn;
            const bool hitlane = mask != 0u;
            const unsigned r_new = ((unsigned)jt << 16) | mask;
#pragma unroll
            for (int c = FILT_KH - 1; c > 0; --c) rec[u][c] = hitlane ? rec[u][c - 1] : rec[u][c];
            rec[u][0] = hitlane ? r_new : rec[u][0];
            cnt[u] += hitlane ? 1 : 0;
        }
    }
#pragma unroll
    for (int u = 0; u < FILT_ATW; ++u) {
        const int ti = ti0 + u;
        if (ti >= ntA) break;  // (wave-uniform)
        const int i = ti * TILE_ROWS + col;
        if (row_ok[u]) {
            const int64_t slot = ((int64_t)g * T1 + i) * 2 + h;
            cand_n[slot] = (uint8_t)(cnt[u] < 255 ? cnt[u] : 255);
            uint4 out = uint4{rec[u][0], rec[u][1], rec[u][2], rec[u][3]};
            *reinterpret_cast<uint4 *>(cand + slot * FILT_KH) = out;
        }
        // a lane whose list overflowed (exact ties, monotone columns), or a norm out of range: the tile's rows get the
        // fp32 pass (the fallback waves of the k_exact_rows launch)
        const unsigned long long ov = __ballot(row_ok[u] && cnt[u] > FILT_KH);
        if (lane == 0) tile_flag[(int64_t)g * ntA + ti] = (ov || force_group || ((a_out >> u) & 1u)) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 pass of ONE flagged A tile (32 rows), one wave, inside the k_exact_rows launch: the arithmetic of
// k_scores_rowmax -- S^T tiles on v_mfma_f32_32x32x2_f32, k ascending (the contract's fma chain), first maximum -- over
// ALL B tiles of the group, with the fp32 unit vectors u = fdiv(v, ||v||) made on the fly from the bf16 means and norms
// (what k_unit_rows_heads would have written: same v, same norm, same correctly rounded division), in the fragment
// order of that kernel: float4 q of lane (col, h) = channels 8q + h, 8q + 2 + h, 8q + 4 + h, 8q + 6 + h of row col.
// Replaces the two launches (k_units_from_means + masked k_scores_rowmax) that ran -- idle -- behind every matching.
// A flagged tile is rare (exact ties / duplicated tokens, norms out of range); when EVERY tile is flagged this sweep is
// the whole similarity on the fp32 pipe with 32 divisions per lane and tile in front of 32 matrix instructions.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 unit_quad(const uint4 &raw, int h, float nr) {
    Pack<bf16_t, 8> pk;
    __builtin_memcpy(&pk, &raw, 16);
    f32x4 u;
    u.x = __fdiv_rn(to_f32(h ? pk.e[1] : pk.e[0]), nr);
    u.y = __fdiv_rn(to_f32(h ? pk.e[3] : pk.e[2]), nr);
    u.z = __fdiv_rn(to_f32(h ? pk.e[5] : pk.e[4]), nr);
    u.w = __fdiv_rn(to_f32(h ? pk.e[7] : pk.e[6]), nr);
    return u;
}

__device__ __forceinline__ void fp32_pass_of_tile(const uint4 *__restrict__ vA, const uint4 *__restrict__ vB,
                                                  const float *__restrict__ normA, const float *__restrict__ normB,
                                                  int g, int ti, int T1, int T2, int ntA, int ntB, int distill_token,
                                                  float *__restrict__ node_max, int *__restrict__ node_idx) {
    const int lane = threadIdx.x & 63;
    const int col = lane & 31, h = lane >> 5;
    const int i = ti * TILE_ROWS + col;
    f32x4 af[8];
    {
        const float ni = normA[(int64_t)g * T1 + (i < T1 ? i : T1 - 1)];
        const uint4 *at = vA + ((int64_t)g * ntA + ti) * 256;
#pragma unroll
        for (int q = 0; q < 8; ++q) af[q] = unit_quad(at[bfrag_index(0, q >> 1, col + 32 * (q & 1))], h, ni);
    }
    RowBest rb = {-INFINITY, 0};
    uint4 braw[8];
    float nj;
    {
        const uint4 *bt = vB + (int64_t)g * ntB * 256;
#pragma unroll
        for (int q = 0; q < 8; ++q) braw[q] = bt[bfrag_index(0, q >> 1, col + 32 * (q & 1))];
        nj = normB[(int64_t)g * T2 + (col < T2 ? col : T2 - 1)];
    }
    for (int jt = 0; jt < ntB; ++jt) {
        uint4 cur[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) cur[q] = braw[q];
        const float cn = nj;
        {   // the next tile's means and norms (the last iteration re-reads its own)
            const int nx = jt + 1 < ntB ? jt + 1 : jt;
            const uint4 *bt = vB + ((int64_t)g * ntB + nx) * 256;
#pragma unroll
            for (int q = 0; q < 8; ++q) braw[q] = bt[bfrag_index(0, q >> 1, col + 32 * (q & 1))];
            const int jn = nx * TILE_ROWS + col;
            nj = normB[(int64_t)g * T2 + (jn < T2 ? jn : T2 - 1)];
        }
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 b = unit_quad(cur[q], h, cn), a = af[q];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc, 0, 0, 0);
        }
        fold_tile(acc, rb, jt, h, T2, distill_token);
    }
    {   // the two lane halves hold the same A row, disjoint B rows: keep the larger, first index on ties
        const float ob = __shfl_xor(rb.best, 32);
        const int oi = __shfl_xor(rb.idx, 32);
        if (ob > rb.best || (ob == rb.best && oi < rb.idx)) {
            rb.best = ob;
            rb.idx = oi;
        }
    }
    if (h == 0 && i < T1) {
        node_max[(int64_t)g * T1 + i] = rb.best;
        node_idx[(int64_t)g * T1 + i] = rb.idx;
    }
}

// ------------------------------------------------------------------------------------------------
// k_exact_rows: blocks [0, exact_blocks): one lane per row of an UNFLAGGED tile -- the recorded columns within the
// window of the row's final approximate maximum, each evaluated with the contract's arithmetic.  Blocks behind them:
// four waves each, one flagged tile per wave on the fp32 matrix pipe (fp32_pass_of_tile; a wave whose tile is not
// flagged leaves at once).  Output: node arrays [n][T1] in the layout k_rank_select reads with nparts = 1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_exact_rows(const uint4 *__restrict__ vA, const uint4 *__restrict__ vB,
                                                    const float *__restrict__ normA, const float *__restrict__ normB,
                                                    int n, int T1, int T2, int D, int ntA, int ntB,
                                                    const CandEntry *__restrict__ cand,
                                                    const uint8_t *__restrict__ cand_n,
                                                    const uint8_t *__restrict__ tile_flag, int exact_blocks,
                                                    int distill_token, float *__restrict__ node_max,
                                                    int *__restrict__ node_idx) {
    if ((int)blockIdx.x >= exact_blocks) {
        const int64_t tile = ((int64_t)blockIdx.x - exact_blocks) * 4 + (threadIdx.x >> 6);
        if (tile >= (int64_t)n * ntA || !tile_flag[tile]) return;  // (wave-uniform)
        const int gf = (int)(tile / ntA);
        fp32_pass_of_tile(vA, vB, normA, normB, gf, (int)(tile - (int64_t)gf * ntA), T1, T2, ntA, ntB, distill_token,
                          node_max, node_idx);
        return;
    }
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (int64_t)n * T1) return;
    const int g = (int)(row / T1), i = (int)(row - (int64_t)g * T1);
    const int ti = i >> 5;
    if (tile_flag[(int64_t)g * ntA + ti]) return;  // (a fallback wave of this launch writes the tile's rows)
    // Everything the row needs from memory is requested in two rounds -- (counts, records, this row's bf16 mean), then
    // (the recorded columns' means and norms) -- instead of one dependent round trip per record and per column: the
    // first form of this kernel spent 48 us per sweep over the columns waiting for them one by one.
    const CandEntry *e0 = cand + row * 2 * FILT_KH;
    const int n0 = min((int)cand_n[row * 2], FILT_KH), n1 = min((int)cand_n[row * 2 + 1], FILT_KH);
    const uint4 q0 = *reinterpret_cast<const uint4 *>(e0), q1 = *reinterpret_cast<const uint4 *>(e0 + FILT_KH);
    const float ni = normA[row];
    uint4 vi[8];  // this row's bf16 mean as it lies in the fragment plane: chunk b = channels 8b .. 8b+7
    {
        const uint4 *at = vA + ((int64_t)g * ntA + ti) * 256;
#pragma unroll
        for (int b = 0; b < 8; ++b) vi[b] = at[bfrag_index(0, b >> 1, (i & 31) + 32 * (b & 1))];
    }
    const float inv_i = __builtin_amdgcn_rcpf(ni);
    // the recorded columns, in record order (half 0 first), the first EX_NC of them in registers
    constexpr int EX_NC = 4;
    int jc[EX_NC], nc = 0, total = 0;
    const unsigned recs[2 * FILT_KH] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int e = 0; e < EX_NC; ++e) jc[e] = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int c = 0; c < FILT_KH; ++c) {
            unsigned mask = (c < (half ? n1 : n0)) ? (recs[half * FILT_KH + c] & 0xFFFFu) : 0u;
            const int jt = (int)(recs[half * FILT_KH + c] >> 16);
            while (mask) {
                const int bit = 31 - __builtin_clz(mask);  // bit 15 - v
                mask &= ~(1u << bit);
                const int v = 15 - bit;
                const int j = jt * TILE_ROWS + (v & 3) + 8 * (v >> 2) + 4 * half;
                if (j >= T2 || (distill_token && j == 0)) continue;  // (absent / protected columns: merge.py:61-62)
#pragma unroll
                for (int e = 0; e < EX_NC; ++e)
                    if (nc == e) jc[e] = j;
                nc += nc < EX_NC ? 1 : 0;
                ++total;
            }
        }
    }
    // approximate cosine of a column (v_dot2_f32_bf16 over the packed means, times 1 / (||v_i|| ||v_j||): inside the
    // filter's error budget) and the contract's value (2 x 64 divisions + the fma chain in k order), from the
    // column's 8 chunks in registers
    typedef __bf16 ex_bf16x2 __attribute__((ext_vector_type(2)));
    auto gather = [&](int j, uint4 (&vj)[8], float &nj) __attribute__((always_inline)) {
        nj = normB[(int64_t)g * T2 + j];
        const uint4 *bt = vB + ((int64_t)g * ntB + (j >> 5)) * 256;
#pragma unroll
        for (int b = 0; b < 8; ++b) vj[b] = bt[bfrag_index(0, b >> 1, (j & 31) + 32 * (b & 1))];
    };
    auto approx = [&](const uint4 (&vj)[8], float nj) __attribute__((always_inline)) -> float {
        float dot = 0.0f;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned ai[4] = {vi[b].x, vi[b].y, vi[b].z, vi[b].w}, bj[4] = {vj[b].x, vj[b].y, vj[b].z, vj[b].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ex_bf16x2 x, y;
                __builtin_memcpy(&x, &ai[e], 4);
                __builtin_memcpy(&y, &bj[e], 4);
                dot = __builtin_amdgcn_fdot2_f32_bf16(x, y, dot, false);
            }
        }
        const float c = dot * inv_i * __builtin_amdgcn_rcpf(nj);
        return c == c ? c : -INFINITY;
    };
    auto exact = [&](const uint4 (&vj)[8], float nj) __attribute__((always_inline)) -> float {
        float acc = 0.0f;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            Pack<bf16_t, 8> pi, pj;
            __builtin_memcpy(&pi, &vi[b], 16);
            __builtin_memcpy(&pj, &vj[b], 16);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (8 * b + e < D) acc = __fmaf_rn(__fdiv_rn(to_f32(pi.e[e]), ni), __fdiv_rn(to_f32(pj.e[e]), nj), acc);
        }
        return acc;
    };
    float best = -INFINITY;
    int bidx = 0;
    bool have = false;
    auto take = [&](int j, float acc) __attribute__((always_inline)) {
        if (acc != acc) return;  // (a NaN never wins the MFMA pass either: `x > NaN` is false)
        // the largest exact score; among equal ones the smallest column (torch.max: first maximum)
        if (!have || acc > best || (acc == best && j < bidx)) {
            best = acc;
            bidx = j;
            have = true;
        }
    };
    // slots 0 and 1 (a row has one recorded column per lane half unless near-ties add more): both fetched at once
    uint4 w0[8], w1[8];
    float nj0, nj1;
    gather(jc[0], w0, nj0);  // (a lane without the slot reads column jc[.] = 0: in range, ignored)
    gather(jc[1], w1, nj1);
    float ca[EX_NC];
    ca[0] = nc > 0 ? approx(w0, nj0) : -INFINITY;
    ca[1] = nc > 1 ? approx(w1, nj1) : -INFINITY;
#pragma unroll
    for (int e = 2; e < EX_NC; ++e) {
        ca[e] = -INFINITY;
        if (__ballot(e < nc)) {  // (wave-uniform: slots nobody fills are skipped)
            uint4 wx[8];
            float njx;
            gather(jc[e], wx, njx);
            const float t = approx(wx, njx);
            ca[e] = e < nc ? t : -INFINITY;
        }
    }
    float amax = ca[0];
#pragma unroll
    for (int e = 1; e < EX_NC; ++e) amax = fmaxf(amax, ca[e]);
    // the better of slots 0 / 1 first: one exact evaluation per row in the common case
    {
        const bool sw = ca[1] > ca[0];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint4 t0 = w0[b], t1 = w1[b];
            w0[b] = sw ? t1 : t0;
            w1[b] = sw ? t0 : t1;
        }
        const float tn = nj0, tc = ca[0];
        const int tj = jc[0];
        nj0 = sw ? nj1 : nj0; nj1 = sw ? tn : nj1;
        ca[0] = sw ? ca[1] : ca[0]; ca[1] = sw ? tc : ca[1];
        jc[0] = sw ? jc[1] : jc[0]; jc[1] = sw ? tj : jc[1];
    }
    {
        const float cx = exact(w0, nj0);
        if (nc > 0 && ca[0] >= amax - FILT_WINDOW) take(jc[0], cx);
    }
    {
        const bool need = nc > 1 && ca[1] >= amax - FILT_WINDOW;
        if (__ballot(need)) {
            const float cx = exact(w1, nj1);
            if (need) take(jc[1], cx);
        }
    }
#pragma unroll
    for (int e = 2; e < EX_NC; ++e) {
        const bool need = e < nc && ca[e] >= amax - FILT_WINDOW;
        if (__ballot(need)) {
            uint4 wx[8];
            float njx;
            gather(jc[e], wx, njx);
            const float cx = exact(wx, njx);
            if (need) take(jc[e], cx);
        }
    }
    if (__ballot(total > EX_NC)) {
        // more recorded columns than slots (near-ties on top of near-ties): the lanes concerned walk their whole
        // lists, column by column
        if (total > EX_NC) {
            best = -INFINITY;
            bidx = 0;
            have = false;
            float am = -INFINITY;
            for (int pass = 0; pass < 2; ++pass)
                for (int half = 0; half < 2; ++half)
                    for (int c = 0; c < (half ? n1 : n0); ++c) {
                        const unsigned r = e0[half * FILT_KH + c];
                        const int jt = (int)(r >> 16);
                        unsigned mask = r & 0xFFFFu;
                        while (mask) {
                            const int bit = 31 - __builtin_clz(mask);
                            mask &= ~(1u << bit);
                            const int v = 15 - bit;
                            const int j = jt * TILE_ROWS + (v & 3) + 8 * (v >> 2) + 4 * half;
                            if (j >= T2 || (distill_token && j == 0)) continue;
                            uint4 wx[8];
                            float njx;
                            gather(j, wx, njx);
                            const float t = approx(wx, njx);
                            if (pass == 0) am = fmaxf(am, t);
                            else if (t >= am - FILT_WINDOW) take(j, exact(wx, njx));
                        }
                    }
        }
    }
    node_max[row] = best;
    node_idx[row] = bidx;
}
