// tome_attn_stream.h -- k_prop_attention_stream: the attention of tome_attn.h as PERSISTENT workgroups that carry the
// K/V pipeline across query blocks.
//
// Why (tools/attn_diag.py, phase stamps of k_prop_attention at 128 x 12 x 1536, MI355X): a 256-query block spends
// 22.9 us in its 23 pipelined tile steps and 7 us around them -- 0.4 us index arithmetic + Q fetch, 2.5 us for the
// first K/V tile to reach LDS (two memory round trips in a row), 0.7 us first tile's softmax, 1.9 us last product,
// normalisation and stores, 1.5 us until the next workgroup starts on the CU (one 8-wave workgroup per CU: nothing
// overlaps these).  The tile loop itself runs at 1.17 PFLOP/s; the block as a whole at 0.85.
//
// Here one workgroup per CU walks items L = blockIdx.x, + gridDim.x, ... (item = (batch, head, segment, query block),
// same XCD-aware numbering as k_prop_attention) and never drains its pipeline between two of them:
//   * the staging stream (tile t+2 requested while tile t+1 is multiplied) runs on into the NEXT item's first two
//     tiles during the current item's last two steps;
//   * the next item's Q rows are requested when the last softmax of the current item has freed its registers;
//   * the step between two items ("switch") = last product, normalisation, stores (left in flight) + stage + ONE
//     barrier + the new item's first-tile softmax;
//   * the overflow guard of the first-pass softmax is settled inside the wave, tile by tile (see fast_step): no
//     workgroup-wide repeat of the block, hence no second code path through the whole item.
// The ring slot of a tile is (tiles streamed so far) & 1 -- a run-time value here (tiles per item may be odd).
// Everything else -- fragment layouts, the first-pass softmax against the first tile's reference point with the
// accumulators started at -m, the overflow guard with the general-path repeat, masked last tile, bias forms, segments
// -- is k_prop_attention's and shares its device functions' contracts (see tome_attn.h).
//
// Needs at least two key tiles (Nk > 64) and eight waves; the host takes k_prop_attention otherwise.
#pragma once
#include "tome_attn.h"

#ifdef ATT_DIAG
// (diagnostic build: the third item of every workgroup stamps its phases)
#define ATS_STAMP(i)                                                                                       \
    do {                                                                                                   \
        if (diag_n == 2 && blockIdx.x < ATT_DIAG_WGS) {                                                    \
            const unsigned long long st_ = __builtin_amdgcn_s_memrealtime();                               \
            if (lane == 0) g_att_stamps[((size_t)blockIdx.x * 8 + wave) * ATT_DIAG_N + (i)] = st_;          \
        }                                                                                                  \
    } while (0)
#else
#define ATS_STAMP(i)
#endif

#define ATS_BIAS_GROUP 4  // tiles per bias load (power of two; 2 * GROUP * 64 words of LDS)
template <typename TX, bool BIAS>
__global__ __launch_bounds__(512, 2) void k_prop_attention_stream(AttnArgs a, int nitems) {
    constexpr int WAVES = 8;
    constexpr int ATT_BM = 32 * WAVES;
    __shared__ __attribute__((aligned(16))) short lds_k[ATT_SLOTS][ATT_BN * ATT_KS];
    __shared__ __attribute__((aligned(16))) short lds_v[ATT_SLOTS][ATT_BN * ATT_VS];
    __shared__ __attribute__((aligned(16))) unsigned lds_bias[2][ATS_BIAS_GROUP * ATT_BN];  // BIAS: (hi | lo << 16) per key, see bias_step()
    __shared__ __attribute__((aligned(16))) short lds_o[WAVES][32 * ATT_KS];  // a wave's output block on its way out

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int col = lane & 31, hf = lane >> 5;
    const int qblocks = (a.N + ATT_BM - 1) / ATT_BM;
    const int nbhs = a.B * a.H * a.nseg;
    const int G = gridDim.x;  // a multiple of 8: item L and item L + G belong to one XCD's group
    const int ntiles = (a.Nk + ATT_BN - 1) / ATT_BN;  // >= 2 (host)
    const int nfull = a.Nk / ATT_BN;
    const bool last_half = (ntiles - 1) * ATT_BN + 32 >= a.Nk;  // the last tile holds at most 32 keys
    const float LOG2E = 1.4426950408889634f;
    const float sl = a.scale * LOG2E;
    // The arguments that only the step between two items reads (strides, base pointers) are read again from the
    // kernel-argument segment there (scalar loads, cached) instead of living in ~50 scalar registers across the tile
    // loop: kept live they push the kernel past the 102-SGPR budget, and the spills go through vector registers.
    typedef const __attribute__((address_space(4))) AttnArgs *AttnArgsK;
    const AttnArgsK kargs = (AttnArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    auto fresh = [&]() __attribute__((always_inline)) -> AttnArgsK {
        AttnArgsK p = kargs;
        asm volatile("" : "+s"(p));  // opaque: the loads behind it are not merged with earlier ones
        return p;
    };
    const int nseg = a.nseg, nheads = a.H, nq = a.N, nk = a.Nk, bias_skip = a.bias_skip;
    const int64_t k_tile = (int64_t)ATT_BN * a.k_sn * 2, v_tile = (int64_t)ATT_BN * a.v_sn * 2;  // bytes per tile

    // ---- item numbering (k_prop_attention's): the query blocks of one (batch, head, segment) -> ids congruent mod 8
    struct Item { int b, h, seg, qb; };
    auto item_ok = [&](int L) -> bool {
        if (L >= nitems) return false;
        const int sq = L >> 3;
        return (sq / qblocks) * 8 + (L & 7) < nbhs;
    };
    auto decode = [&](int L) -> Item {
        const int sq = L >> 3;
        const int bhs = (sq / qblocks) * 8 + (L & 7);
        Item it;
        it.qb = sq % qblocks;
        const int bh = bhs / nseg;
        it.seg = bhs - bh * nseg;
        it.b = bh / nheads;
        it.h = bh % nheads;
        return it;
    };

    // ---- staging: thread -> row r0, 16-byte column c0 of the 64 x 64 K and V tiles; byte offsets from the tile's
    // first row are per-thread constants (the rows of the partly filled last tile clamped to its last key: no bounds
    // branch, those keys' weights are masked)
    const int r0 = tid >> 3, c0 = tid & 7;
    const int rp = min(r0, max(nk - nfull * ATT_BN, 1) - 1);
    // (32-bit: 64 rows of at most 2^24 elements -- the host checks the strides)
    const unsigned koff = (unsigned)(((int64_t)r0 * a.k_sn + 8 * c0) * 2), koff_p = (unsigned)(((int64_t)rp * a.k_sn + 8 * c0) * 2);
    const unsigned voff = (unsigned)(((int64_t)r0 * a.v_sn + 8 * c0) * 2), voff_p = (unsigned)(((int64_t)rp * a.v_sn + 8 * c0) * 2);
    uint4 kreg, vreg;
    unsigned breg = 0u;
    int r_sub = 0, b_par = 0, b_off = 0;  // BIAS: the staged tile's place in its group, the group buffer in use
    // The stream: the next tile to request is tile s_t of item s_item and starts at s_kt / s_vt (bias row: s_ls).
    // n_q / n_o / n_qb: where that item's query block and output block start -- once the stream has moved on to the
    // item after the one being computed (two steps before the switch) these describe the computing side's NEXT item.
    const char *s_kt = nullptr, *s_vt = nullptr;
    const float *s_ls = nullptr;
    int s_t = 0, s_item = 0;
    const short *n_q = nullptr;
    short *n_o = nullptr;
    int n_qb = 0;
    auto stream_set = [&](int L) __attribute__((always_inline)) {
        const Item it = decode(L);
        const AttnArgsK A = fresh();
        s_kt = reinterpret_cast<const char *>(reinterpret_cast<const short *>(A->k) + it.b * A->k_sb + it.h * A->k_sh + it.seg * A->k_seg);
        s_vt = reinterpret_cast<const char *>(reinterpret_cast<const short *>(A->v) + it.b * A->v_sb + it.h * A->v_sh + it.seg * A->v_seg);
        if (BIAS) s_ls = A->log_size + it.b * A->ls_sb + it.seg * A->ls_seg;
        n_q = reinterpret_cast<const short *>(A->q) + it.b * A->q_sb + it.h * A->q_sh + (int64_t)it.qb * ATT_BM * A->q_sn;
        n_o = reinterpret_cast<short *>(A->out) + it.b * A->o_sb + it.h * A->o_sh + it.seg * A->o_seg + (int64_t)it.qb * ATT_BM * A->o_sn;
        n_qb = it.qb;
        s_item = L;
        s_t = 0;
    };
    // request the stream's next tile and move on (past the last item the stream repeats its last tile: valid memory,
    // never used)
    auto stream_load = [&]() __attribute__((always_inline)) {
        const bool part = s_t >= nfull;  // wave-uniform
        kreg = *reinterpret_cast<const uint4 *>(s_kt + (part ? koff_p : koff));
        vreg = *reinterpret_cast<const uint4 *>(s_vt + (part ? voff_p : voff));
        // BIAS: the per-key term of ATS_BIAS_GROUP tiles at a time, one key per thread -- staging it tile by tile
        // put ~25 instructions per step on wave 0 alone, and every wave waits for the slowest at the barrier
        // (measured: 10 % of the kernel)
        r_sub = s_t & (ATS_BIAS_GROUP - 1);
        if (BIAS && r_sub == 0 && tid < ATS_BIAS_GROUP * ATT_BN) {
            const int key = min(s_t * ATT_BN + tid, nk - 1);
            // log2-domain bias as two 16-bit terms hi + lo (what is left is below 2^-15 of the bias for bf16)
            const float bv = key >= bias_skip ? s_ls[key - bias_skip] * LOG2E : 0.0f;
            const short bh = att_bits<TX>(bv);
            TX th;
            __builtin_memcpy(&th, &bh, 2);
            const short bl = att_bits<TX>(bv - to_f32(th));
            breg = (unsigned)(unsigned short)bh | ((unsigned)(unsigned short)bl << 16);
        }
        if (++s_t == ntiles) {
            if (item_ok(s_item + G)) stream_set(s_item + G);
            else s_t = ntiles - 1;
        } else {
            s_kt += k_tile;
            s_vt += v_tile;
        }
    };
    short *const wk = &lds_k[0][0] + r0 * ATT_KS + 8 * c0, *const wv = &lds_v[0][0] + r0 * ATT_VS + 8 * c0;
    auto stage_write = [&](int S) __attribute__((always_inline)) {
        *reinterpret_cast<uint4 *>(wk + S * ATT_BN * ATT_KS) = kreg;
        *reinterpret_cast<uint4 *>(wv + S * ATT_BN * ATT_VS) = vreg;
        if (BIAS) {  // (wave-uniform) a tile that opens a group brings the group's bias words along
            if (r_sub == 0) {
                b_par ^= 1;
                if (tid < ATS_BIAS_GROUP * ATT_BN) lds_bias[b_par][tid] = breg;
            }
            b_off = b_par * (ATS_BIAS_GROUP * ATT_BN) + r_sub * ATT_BN;  // where the tile now in slot S finds its words
        }
    };

    // ---- per-item state of the computing side
    const int lrow0 = wave * 32 + col;  // this lane's query row inside a block
    const int q_sn = (int)a.q_sn, o_sn = (int)a.o_sn;  // (< 2^24: host)
    int c_qb = 0;             // the item's query block
    short *c_o = nullptr;     // where its output block starts
    int qrow = 0;
    bool active = false;
    float bfac = 1.0f;
    att_s16x8 qf[4];
    uint4 qraw[4];
    // request the Q rows of the stream's item (the computing side's NEXT item) and note where its output goes --
    // before the stream moves on; item_take then makes it the current item
    int t_qb = 0;
    short *t_o = nullptr;
    auto q_request = [&]() __attribute__((always_inline)) {
        t_qb = n_qb;
        t_o = n_o;
        const int lrow = max(min(lrow0, nq - 1 - t_qb * ATT_BM), 0);
        const short *qp = n_q + lrow * q_sn + 8 * hf;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qraw[ks] = *reinterpret_cast<const uint4 *>(qp + 16 * ks);
    };
    auto item_take = [&]() __attribute__((always_inline)) {
        c_qb = t_qb;
        c_o = t_o;
        qrow = c_qb * ATT_BM + lrow0;
        active = c_qb * ATT_BM + wave * 32 < nq;  // wave-uniform
        bfac = (bias_skip && qrow == 0) ? 0.0f : 1.0f;
    };
    auto q_convert = [&]() __attribute__((always_inline)) {  // q * scale * log2(e), rounded once to the 16-bit format
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            att_s16x8 raw;
            __builtin_memcpy(&raw, &qraw[ks], 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                TX tq;
                const short r = raw[e];
                __builtin_memcpy(&tq, &r, 2);
                qf[ks][e] = att_bits<TX>(to_f32(tq) * sl);
            }
        }
    };

    att_f32x16 o0, o1, negm, s0, s1;
    float m_run = -INFINITY, l_run = 0.0f;
#ifdef ATT_DIAG
    int diag_n = 0;
    unsigned long long diag_wait_stage = 0, diag_wait_barrier = 0;
#endif
    auto item_reset = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < 16; ++v) o0[v] = o1[v] = negm[v] = 0.0f;
        m_run = -INFINITY;
        l_run = 0.0f;
    };

    // ---- fragments and products (tome_attn.h's layouts; the slot is a run-time value)
    typedef __attribute__((address_space(3))) att_s16x4 *lds_s16x4_p;
    const short *const vbase = &lds_v[0][0] + (4 * hf + ((lane & 15) >> 2)) * ATT_VS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const short *const kbase = &lds_k[0][0] + col * ATT_KS + 8 * hf;
    att_s16x4 vfr[2][2][4];
    auto v_fragments = [&](int S) __attribute__((always_inline)) {
        const short *vs = vbase + S * ATT_BN * ATT_VS;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const short *va = vs + (32 * kb + 16 * p) * ATT_VS;
                vfr[kb][p][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va));
                vfr[kb][p][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 8 * ATT_VS));
                vfr[kb][p][2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 32));
                vfr[kb][p][3] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 8 * ATT_VS + 32));
            }
    };
    // BIAS: the per-key term rides on the matrix pipe as a fifth k-step of the score product -- K gets two more
    // "channels" (hi, lo of the bias), Q gets bfac in both -- instead of 32 fma + 8 wide LDS reads per tile on the
    // vector pipe; the lanes of the upper half (channels 8..15 of that k-step) hold zeros
    auto bias_step = [&](int S) __attribute__((always_inline)) {
        const unsigned w0 = lds_bias[0][b_off + col], w1 = lds_bias[0][b_off + 32 + col];
        const unsigned one = (unsigned)(unsigned short)att_bits<TX>(bfac);
        const unsigned qw = hf ? 0u : (one | (one << 16));
        const uint4 q5 = make_uint4(qw, 0u, 0u, 0u), k0 = make_uint4(hf ? 0u : w0, 0u, 0u, 0u), k1 = make_uint4(hf ? 0u : w1, 0u, 0u, 0u);
        att_s16x8 fq, f0, f1;
        __builtin_memcpy(&fq, &q5, 16);
        __builtin_memcpy(&f0, &k0, 16);
        __builtin_memcpy(&f1, &k1, 16);
        s0 = AttMfma<TX>::run(f0, fq, s0);
        s1 = AttMfma<TX>::run(f1, fq, s1);
    };
    auto scores = [&](int S, const att_f32x16 &cinit, bool first_half_only) __attribute__((always_inline)) {
        const short *ks0 = kbase + S * ATT_BN * ATT_KS;
        s0 = cinit;
        s1 = cinit;
        if (BIAS) bias_step(S);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            s0 = AttMfma<TX>::run(*reinterpret_cast<const att_s16x8 *>(ks0 + 16 * ks), qf[ks], s0);
        if (!first_half_only) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s1 = AttMfma<TX>::run(*reinterpret_cast<const att_s16x8 *>(ks0 + 32 * ATT_KS + 16 * ks), qf[ks], s1);
        }
    };
    att_s16x8 pf[2][2];
    auto pack_p = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pf[0][p][e] = att_bits<TX>(s0[8 * p + e]);
                pf[1][p][e] = att_bits<TX>(s1[8 * p + e]);
            }
    };
    // general softmax of tile t from scores that start at zero: range mask, maximum, rescale of O and l
    auto general_softmax = [&](int t, int S) __attribute__((always_inline)) {
        att_f32x16 zero;
#pragma unroll
        for (int v = 0; v < 16; ++v) zero[v] = 0.0f;
        scores(S, zero, false);
        const int key0 = t * ATT_BN + 4 * hf;
        float mt = -INFINITY;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int key = key0 + (v & 3) + 8 * (v >> 2);
            s0[v] = key < nk ? s0[v] : -INFINITY;
            s1[v] = key + 32 < nk ? s1[v] : -INFINITY;
            mt = fmaxf(mt, fmaxf(s0[v], s1[v]));
        }
        {
            const unsigned mb = __float_as_uint(mt);
            const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_new = fmaxf(m_run, mt);  // finite: every tile holds at least one key in range
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float lsum = 0.0f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            s0[v] = __builtin_amdgcn_exp2f(s0[v] - m_new);
            s1[v] = __builtin_amdgcn_exp2f(s1[v] - m_new);
            lsum += s0[v] + s1[v];
        }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            o0[v] *= alpha;
            o1[v] *= alpha;
            negm[v] = -m_new;
        }
        pack_p();
    };
    // the first tile of an item (full: an item has at least two tiles): nothing to rescale, nothing to mask --
    // m = the tile's row maximum, P = exp2(s - m), l = the row sum
    auto first_softmax = [&](int S) __attribute__((always_inline)) {
        att_f32x16 zero;
#pragma unroll
        for (int v = 0; v < 16; ++v) zero[v] = 0.0f;
        scores(S, zero, false);
        float mt = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int v = 1; v < 16; ++v) mt = fmaxf(mt, fmaxf(s0[v], s1[v]));
        {
            const unsigned mb = __float_as_uint(mt);
            const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        float lsum = 0.0f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            s0[v] = __builtin_amdgcn_exp2f(s0[v] - mt);
            s1[v] = __builtin_amdgcn_exp2f(s1[v] - mt);
            lsum += s0[v] + s1[v];
        }
        l_run = lsum;
        m_run = mt;
#pragma unroll
        for (int v = 0; v < 16; ++v) negm[v] = -mt;
        pack_p();
    };
    auto pv = [&](bool first_half_only) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && first_half_only) break;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                att_s16x8 vf0, vf1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vf0[e] = vfr[kb][p][0][e];
                    vf0[4 + e] = vfr[kb][p][1][e];
                    vf1[e] = vfr[kb][p][2][e];
                    vf1[4 + e] = vfr[kb][p][3][e];
                }
                o0 = AttMfma<TX>::run(vf0, pf[kb][p], o0);
                o1 = AttMfma<TX>::run(vf1, pf[kb][p], o1);
            }
        }
    };
    // one pipelined step inside an item: tile tn (in the staging registers) -> slot S, the stream's next tile
    // requested, ONE barrier, S(tn) and O += V(tn-1) P(tn-1), then P(tn) against the item's reference point
    auto fast_step = [&](int S, int tn, bool masked) __attribute__((always_inline)) {
#ifdef ATT_DIAG
        // (diagnostic build: how long does a step wait for the tile it requested one step ago, and at the barrier?)
        unsigned long long d0_ = 0, d1_ = 0, d2_ = 0;
        if (diag_n == 2) d0_ = __builtin_amdgcn_s_memrealtime();
#endif
        stage_write(S);
#ifdef ATT_DIAG
        if (diag_n == 2) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            d1_ = __builtin_amdgcn_s_memrealtime();
        }
#endif
        stream_load();
        __syncthreads();
#ifdef ATT_DIAG
        if (diag_n == 2) {
            d2_ = __builtin_amdgcn_s_memrealtime();
            diag_wait_stage += d1_ - d0_;
            diag_wait_barrier += d2_ - d1_;
        }
#endif
        scores(S, negm, masked && last_half);
        pv(false);
        v_fragments(S);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int v = 0; v < 16; ++v) s0[v] = __builtin_amdgcn_exp2f(s0[v]);
#pragma unroll
        for (int v = 0; v < 16; ++v) s1[v] = __builtin_amdgcn_exp2f(s1[v]);
        if (masked) {  // the partly filled last tile: keys past the end weigh nothing
            const int key0 = tn * ATT_BN + 4 * hf;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int key = key0 + (v & 3) + 8 * (v >> 2);
                s0[v] = key < nk ? s0[v] : 0.0f;
                s1[v] = key + 32 < nk ? s1[v] : 0.0f;
            }
        }
        float c0s = att_add(s0[0], s0[4]), c1s = att_add(s0[1], s0[5]), c2s = att_add(s0[2], s0[6]),
              c3s = att_add(s0[3], s0[7]);
#pragma unroll
        for (int v = 8; v < 16; v += 4) {
            c0s = att_add(c0s, s0[v]);     c1s = att_add(c1s, s0[v + 1]);
            c2s = att_add(c2s, s0[v + 2]); c3s = att_add(c3s, s0[v + 3]);
        }
#pragma unroll
        for (int v = 0; v < 16; v += 4) {
            c0s = att_add(c0s, s1[v]);     c1s = att_add(c1s, s1[v + 1]);
            c2s = att_add(c2s, s1[v + 2]); c3s = att_add(c3s, s1[v + 3]);
        }
        const float lsum = att_add(att_add(c0s, c1s), att_add(c2s, c3s));
        pack_p();
        __builtin_amdgcn_s_setprio(0);
        const float l_before = l_run;
        l_run += lsum;
        // Overflow guard, at the very end of the step (a branch in the middle of it keeps the compiler from
        // interleaving the conversions with what follows: +10 % per step, measured): a row sum beyond the 16-bit
        // format's range (or inf / NaN) says the reference point lags the scores by more than the format allows.
        // The WAVE then takes this tile again on the general path -- its K tile is still in slot S, O and l hold the
        // tiles before it -- which moves the reference point to the new maximum; the following tiles continue on
        // the fast path against it.  (Wave-uniform branch, adversarial inputs only; k_prop_attention repeats the
        // whole block instead.)
#ifndef ATS_NOGUARD
        if (__builtin_amdgcn_ballot_w64(!(lsum <= AttLimit<TX>::value)) != 0) {
            l_run = l_before;
            general_softmax(tn, S);
        }
#endif
    };
    auto helper_step = [&](int S) __attribute__((always_inline)) {  // a wave without queries stages and meets the barrier
        stage_write(S);
        stream_load();
        __syncthreads();
    };
    // out[b, q, h*64 + d] = O^T[d][q] / l (tome_attn.h's store: one 16-byte store per lane and channel group)
    // out[b, q, h*64 + d] = O^T[d][q] / l.  A lane holds 4 x 8 channels of ONE query row; stored from there, one store
    // instruction touches 32 rows x 2 x 16 bytes and the eight waves' stores queue up behind each other (measured:
    // waves 4-7 spend 2.2 us here, waves 0-3 0.7).  So the wave's 32 x 64 block goes through a region of LDS of its
    // own (row stride 144 B, the K tile's conflict-free stride) and leaves as whole 128-byte rows: 8 rows per store
    // instruction.  LDS operations of one wave execute in order: no barrier.
    auto store_item = [&]() __attribute__((always_inline)) {
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = __builtin_amdgcn_rcpf(l_tot);  // (1 ulp; the result is rounded to 8 or 11 bits)
        short *ow = &lds_o[0][0] + wave * (32 * ATT_KS);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            att_s16x4 w0, w1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w0[e] = att_bits<TX>(o0[4 * g + e] * inv);
                w1[e] = att_bits<TX>(o1[4 * g + e] * inv);
            }
            unsigned a2[2], b2[2];
            __builtin_memcpy(a2, &w0, 8);
            __builtin_memcpy(b2, &w1, 8);
            // (tome_attn.h's exchange: the lower lanes end up with channels 8g .. 8g+7, the upper with 32+8g .. 32+8g+7)
            const auto s0w = __builtin_amdgcn_permlane32_swap(a2[0], b2[0], false, false);
            const auto s1w = __builtin_amdgcn_permlane32_swap(a2[1], b2[1], false, false);
            *reinterpret_cast<uint4 *>(ow + col * ATT_KS + 32 * hf + 8 * g) = uint4{s0w[0], s1w[0], s0w[1], s1w[1]};
        }
        const int rows = nq - c_qb * ATT_BM - wave * 32;  // rows of this wave inside the sequence (>= 1: active)
        short *op = c_o + (wave * 32 + (lane >> 3)) * o_sn + 8 * (lane & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 row16 = *reinterpret_cast<const uint4 *>(ow + ((lane >> 3) + 8 * i) * ATT_KS + 8 * (lane & 7));
            if ((lane >> 3) + 8 * i < rows) *reinterpret_cast<uint4 *>(op + 8 * i * o_sn) = row16;
        }
    };
    // first item of a workgroup: nothing is in flight yet
    auto cold_start = [&](int L) __attribute__((always_inline)) {
        stream_set(L);
        q_request();    // Q rows requested first,
        item_take();
        item_reset();
        stream_load();  // tile 0 behind them, before anything waits
        q_convert();
        stage_write(0);
        stream_load();  // tile 1
        __syncthreads();
        if (active) {
            first_softmax(0);
            v_fragments(0);
        }
    };

    int item = blockIdx.x;
    if (!item_ok(item)) return;  // (then so is every later item of this workgroup)
    cold_start(item);
    int par = 0;  // ring slot of the current item's tile 0
    for (;;) {
        ATS_STAMP(0);  // tile loop starts
        // ---- steps inside the item: tiles 1 .. ntiles-1 (tile tn -> slot (par + tn) & 1)
        if (active) {
            int tn = 1;
            for (; tn < nfull; ++tn) fast_step((par + tn) & 1, tn, false);
            if (tn < ntiles) fast_step((par + tn) & 1, tn, true);  // the partly filled last tile
        } else {
            for (int tn = 1; tn < ntiles; ++tn) helper_step((par + tn) & 1);
        }
        ATS_STAMP(1);  // tile loop done
#ifdef ATT_DIAG
        if (diag_n == 2 && blockIdx.x < ATT_DIAG_WGS && lane == 0) {
            g_att_stamps[((size_t)blockIdx.x * 8 + wave) * ATT_DIAG_N + 6] = diag_wait_stage;
            g_att_stamps[((size_t)blockIdx.x * 8 + wave) * ATT_DIAG_N + 7] = diag_wait_barrier;
        }
#endif
        // ---- switch: the staging registers hold the next item's tile 0 (if there is a next item).  Order: everything
        // that READS memory for the next item is issued first (its tile 0 to LDS, its Q rows, its tile 1), then the
        // current item's last product, normalisation and stores -- a wait for any of those loads then never waits
        // for the stores behind them (the vector-memory counter runs in order).  The two arms of `active` each hold
        // the barrier, so that the wait counts in front of it stay exact on both.
        const int next = item + G;
        const bool has_next = item_ok(next);  // workgroup-uniform
        const int Sn = (par + ntiles) & 1;
        if (has_next) {
            stage_write(Sn);
            q_request();    // (the stream is at the next item since two steps)
            stream_load();  // the next item's tile 1
        }
        const bool was_active = active;
        if (was_active) {
            pv(last_half);
            store_item();
            ATS_STAMP(2);  // stores issued
            if (!has_next) return;
            __syncthreads();
            ATS_STAMP(3);  // behind the barrier
        } else {
            if (!has_next) return;
            __syncthreads();
        }
        item = next;
        par = Sn;
        item_take();
        item_reset();
        q_convert();
        ATS_STAMP(4);  // Q fragment ready
        if (active) {
            first_softmax(Sn);
            v_fragments(Sn);
        }
        ATS_STAMP(5);  // first tile's weights
#ifdef ATT_DIAG
        ++diag_n;
#endif
    }
}
