// tome_attn_resident.h -- the proportional attention of the ToMe patches for SHORT key sequences (at most 224 keys per
// (batch, head, segment)): TimeSformer's spatial attention (`ToMeAttention.forward`, tome/patch/timesformer.py:66-78:
// 1 + p <= 197 tokens per frame) and the per-frame stage of Motionformer's trajectory attention
// (`ToMeTrajectoryAttention.forward`, tome/patch/motionformer.py:98-121: every query against the <= 196 keys of ONE
// frame at a time, softmax per frame).  Part of the single translation unit csrc/tome_kernels.hip.
//
// Why a kernel of its own.  k_prop_attention[_stream] (tome_attn.h) streams K/V through a two-slot LDS ring, one
// barrier per 64-key tile, and pays ~3.3 us per 256-query block around the tile loop.  With 197 keys that loop is four
// steps (the last for 5 keys): 9.1 us per block at 512 x 12 x 197 = 279 TFLOP/s, a tenth of the matrix peak -- and for
// Motionformer every (query block, frame) pair re-stages the frame's keys (8 frames x 7 query blocks per head).
// Here ONE workgroup owns one (batch, head, segment): its whole K and V (<= 224 rows, 75 KB of LDS, so two workgroups
// share a CU and one's staging runs under the other's arithmetic) are written to LDS once, ONE barrier, and then the
// eight waves walk the query tiles of 32 rows with no further synchronisation -- TimeSformer: one tile per wave;
// Motionformer: all 1 + S*F queries of the clip against the resident frame, 50 tiles per workgroup.
// The arithmetic is k_prop_attention's: S^T = K Q~^T and O^T += V^T P^T on v_mfma_f32_32x32x16 (Q~ = q * scale * log2 e
// rounded once to the 16-bit format), first tile with the general softmax, later tiles with score accumulators that
// START at -m_run (+ the key's log(size)), a row sum that doubles as overflow guard; a wave whose guard trips repeats
// ITS query tile on the general path (reference point following the maximum tile by tile) -- the keys are resident, so
// nobody else is involved.  Roofline of the TimeSformer shape: 100 KB of q/k/v/out per 9.9 MFLOP item = HBM bound at
// ~590 TFLOP/s-equivalent; of the Motionformer shape: 175 FLOP/B, matrix pipe and HBM about level.
#pragma once

#define RES_ROWS 224  // K / V rows resident per workgroup: 3 full 64-key tiles + the first half of a fourth
#ifndef RES_VS
#define RES_VS ATT_VS  // V row stride in LDS (elements)
#endif

template <typename TX, bool BIAS>
__global__ __launch_bounds__(512, 4) void k_resident_attention(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) short lds_k[RES_ROWS * ATT_KS];  // 32256 B
    __shared__ __attribute__((aligned(16))) short lds_v[RES_ROWS * RES_VS];  // 43008 B
    // A FIFTH k-step carries what would otherwise be vector work on 16 accumulator registers per block:
    //   channels 2, 3: K side 1, Q side -m_ref split into two terms hi + lo of the 16-bit format (residual < 2^-15 |m|:
    //              a common factor of at most 2^(1e-4) between the first block's weights and the later ones, far below
    //              the 2^-9 the weights are rounded to) -- the score accumulators start at literal 0 and still deliver
    //              s - m_ref (no 16 moves per block, no 16 registers holding the start values);
    //   channels 0, 1 (BIAS): K side the key's log(size) * log2(e) split into two 16-bit terms hi + lo (residual < 2^-15
    //              of the bias for bf16, 2^-21 for fp16), Q side bfac (0 for TimeSformer's unbiased class query, else 1).
    // The K-side fragment of a key is a 16-byte row of lds_kb (plain: bias terms 0) -- a constant register fragment for
    // the plain form would be four more registers live through the block loop.
    __shared__ __attribute__((aligned(16))) short lds_kb[RES_ROWS * 8];  // 3584 B -> 78848 B per workgroup

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hf = lane >> 5;
    ATT_STAMP(0);  // entry (diagnostic build only: tools/attn_diag.py)
    // XCD-aware item numbering: the nseg segments of one (batch, head) read the same Q rows -> same XCD (workgroup L
    // runs on XCD L % 8), one after the other
    const int L = blockIdx.x;
    const int xcd = L & 7, sq = L >> 3;
    const int bh = (sq / a.nseg) * 8 + xcd, seg = sq % a.nseg;
    if (bh >= a.B * a.H) return;
    const int b = bh / a.H, h = bh % a.H;
    const short *qp = reinterpret_cast<const short *>(a.q) + b * a.q_sb + h * a.q_sh;
    const short *kp = reinterpret_cast<const short *>(a.k) + b * a.k_sb + h * a.k_sh + seg * a.k_seg;
    const short *vp = reinterpret_cast<const short *>(a.v) + b * a.v_sb + h * a.v_sh + seg * a.v_seg;
    const float LOG2E = 1.4426950408889634f;
    const int nk = a.Nk;
    const int ntq = (a.N + 31) >> 5, nblk = (nk + 31) >> 5;  // query tiles and key blocks of 32

    // the first query tile's rows are requested before the staging: their latency runs under it
    uint4 qraw[4];
    {
        const int qload = min(wave * 32 + col, a.N - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qraw[ks] = *reinterpret_cast<const uint4 *>(qp + ((unsigned)qload * (unsigned)a.q_sn + 16u * ks + 8u * hf));
    }
    // ---- K, V (and the bias) of the item -> LDS, rows past the last key zeroed (a weight of 0 meets a finite value)
    {
        constexpr int NCH = RES_ROWS * 8, PER = (NCH + 511) / 512;  // 16-byte chunks per tensor, per thread
        uint4 kr[PER], vr[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + 512 * i, key = idx >> 3, c = idx & 7;
            kr[i] = vr[i] = uint4{0, 0, 0, 0};
            if (key < nk) {
                kr[i] = *reinterpret_cast<const uint4 *>(kp + ((unsigned)key * (unsigned)a.k_sn + 8u * c));
                vr[i] = *reinterpret_cast<const uint4 *>(vp + ((unsigned)key * (unsigned)a.v_sn + 8u * c));
            }
        }
        float br = 0.0f;
        if (BIAS && tid < RES_ROWS) {
            const float *lsp = a.log_size + b * a.ls_sb + seg * a.ls_seg;
            br = (tid < nk && tid >= a.bias_skip) ? lsp[tid - a.bias_skip] * LOG2E : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + 512 * i, key = idx >> 3, c = idx & 7;
            if (idx < NCH) {
                *reinterpret_cast<uint4 *>(lds_k + key * ATT_KS + 8 * c) = kr[i];
                *reinterpret_cast<uint4 *>(lds_v + key * RES_VS + 8 * c) = vr[i];
            }
        }
        if (tid < RES_ROWS) {
            const float hi = to_f32(from_f32<TX>(br));
            att_s16x8 row;
#pragma unroll
            for (int e = 0; e < 8; ++e) row[e] = 0;
            row[0] = att_bits<TX>(hi);
            row[1] = att_bits<TX>(br - hi);
            row[2] = row[3] = att_bits<TX>(1.0f);
            *reinterpret_cast<att_s16x8 *>(lds_kb + tid * 8) = row;
        }
    }
    ATT_STAMP(1);  // staging loads consumed, LDS written
    __syncthreads();  // the only barrier: from here on the waves are on their own
    ATT_STAMP(2);  // K / V resident

    typedef __attribute__((address_space(3))) att_s16x4 *lds_s16x4_p;
    const short *const kbase = lds_k + col * ATT_KS + 8 * hf;
    const short *const vbase = lds_v + (4 * hf + ((lane & 15) >> 2)) * RES_VS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const float sl = a.scale * LOG2E;

    for (int qt = wave; qt < ntq; qt += 8) {
        // (block 0's index as a value the compiler cannot see through: with a literal 0 the K fragments of block 0 -- the
        // same LDS reads for every query tile -- are hoisted out of this loop into 28 registers that the loop body
        // then has to spill)
        int first_blk, last_blk = nblk - 1;
        asm volatile("s_mov_b32 %0, 0" : "=s"(first_blk));
        asm volatile("" : "+s"(last_blk));  // (likewise: the eight V-fragment addresses of the last block)
        const int qrow = qt * 32 + col;
        // TimeSformer form (bias_skip): key 0 and query 0 carry no bias -- the class query's lane multiplies it by 0
        const float bfac = (a.bias_skip && qrow == 0) ? 0.0f : 1.0f;
        // Q side of the fifth k-step (see the top of the kernel); the upper lane half multiplies the K side by zeros
        att_s16x8 qx;
#pragma unroll
        for (int e = 0; e < 8; ++e) qx[e] = 0;
        if (BIAS && hf == 0) qx[0] = qx[1] = att_bits<TX>(bfac);
        ATT_STAMP(3);  // Q~ in registers
        att_s16x8 qf[4];  // q * scale * log2(e), rounded once to the 16-bit format
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

        // The keys are walked in BLOCKS of 32 (one accumulator of scores at a time: with 64-key steps the live registers
        // -- two score accumulators, two output accumulators, Q~, the packed weights -- do not fit the 128 of four waves
        // per SIMD).  Register v of lane l <-> key 32 blk + (v & 3) + 8 (v >> 2) + 4 (l >> 5) of query l & 31.
        att_f32x16 o0, o1, sc;
        float m_run, l_run;
        att_s16x8 pf[2];           // the block's weights in the 16-bit format: B operand of O^T += V^T P^T
        // scores of block blk minus the reference point in qx[2] + qx[3] (0 while the general softmax looks for one)
        auto scores = [&](int blk) __attribute__((always_inline)) {
#pragma unroll
            for (int v = 0; v < 16; ++v) sc[v] = 0.0f;
            const short *kt = kbase + blk * 32 * ATT_KS;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) sc = AttMfma<TX>::run(*reinterpret_cast<const att_s16x8 *>(kt + 16 * ks), qf[ks], sc);
            sc = AttMfma<TX>::run(*reinterpret_cast<const att_s16x8 *>(lds_kb + (blk * 32 + col) * 8), qx, sc);
        };
        auto pack_p = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[p][e] = att_bits<TX>(sc[8 * p + e]);
        };
        // O^T += V(blk)^T P(blk)^T; the V^T fragments are read where they are used (ds_read_b64_tr_b16: 4 consecutive keys
        // of one channel per lane)
        auto pv = [&](int blk) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const short *va = vbase + (blk * 32 + 16 * p) * RES_VS;
                const att_s16x4 f0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va));
                const att_s16x4 f1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 8 * RES_VS));
                const att_s16x4 f2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 32));
                const att_s16x4 f3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 8 * RES_VS + 32));
                att_s16x8 vf0, vf1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vf0[e] = f0[e];
                    vf0[4 + e] = f1[e];
                    vf1[e] = f2[e];
                    vf1[4 + e] = f3[e];
                }
                o0 = AttMfma<TX>::run(vf0, pf[p], o0);
                o1 = AttMfma<TX>::run(vf1, pf[p], o1);
            }
        };
        // general softmax of a block (scores from zero): range mask, maximum, rescale of O and l
        auto general_softmax = [&](int blk) __attribute__((always_inline)) {
            qx[2] = qx[3] = 0;
            scores(blk);
            const int key0 = blk * 32 + 4 * hf;
            float mt = -INFINITY;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                sc[v] = key0 + (v & 3) + 8 * (v >> 2) < nk ? sc[v] : -INFINITY;
                mt = fmaxf(mt, sc[v]);
            }
            {
                const unsigned mb = __float_as_uint(mt);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const float m_new = fmaxf(m_run, mt);  // finite: every block holds at least one key in range
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float lsum = 0.0f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                sc[v] = __builtin_amdgcn_exp2f(sc[v] - m_new);
                lsum += sc[v];
            }
            l_run = l_run * alpha + lsum;
            m_run = m_new;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                o0[v] *= alpha;
                o1[v] *= alpha;
            }
            if (hf == 0) {  // -m_ref for the blocks to come, through the matrix pipe
                const float hi = to_f32(from_f32<TX>(-m_new));
                qx[2] = att_bits<TX>(hi);
                qx[3] = att_bits<TX>(-m_new - hi);
            }
            pack_p();
        };
        auto reset = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int v = 0; v < 16; ++v) o0[v] = o1[v] = 0.0f;
            m_run = -INFINITY;
            l_run = 0.0f;
        };
        // ---- the fast form: first block with the general softmax, every later block's scores arrive as s - m_ref (+ bias)
        bool bad = false;
        reset();
        general_softmax(first_blk);
        for (int blk = 1; blk < nblk; ++blk) {
            pv(blk - 1);
            scores(blk);
#pragma unroll
            for (int v = 0; v < 16; ++v) sc[v] = __builtin_amdgcn_exp2f(sc[v]);
            if ((blk + 1) * 32 > nk) {  // the partly filled last block: keys past the end weigh nothing
                const int key0 = blk * 32 + 4 * hf;
#pragma unroll
                for (int v = 0; v < 16; ++v) sc[v] = key0 + (v & 3) + 8 * (v >> 2) < nk ? sc[v] : 0.0f;
            }
            float c0s = att_add(sc[0], sc[4]), c1s = att_add(sc[1], sc[5]), c2s = att_add(sc[2], sc[6]),
                  c3s = att_add(sc[3], sc[7]);
#pragma unroll
            for (int v = 8; v < 16; v += 4) {
                c0s = att_add(c0s, sc[v]);     c1s = att_add(c1s, sc[v + 1]);
                c2s = att_add(c2s, sc[v + 2]); c3s = att_add(c3s, sc[v + 3]);
            }
            const float lsum = att_add(att_add(c0s, c1s), att_add(c2s, c3s));
            pack_p();
            l_run += lsum;
            bad = bad || !(lsum <= AttLimit<TX>::value);  // inf / NaN / too large: this pass is void
        }
        pv(last_blk);
        ATT_STAMP(4);  // block loop done
        // ---- the guard tripped in this wave (weights beyond the 16-bit format's range: the reference point fixed by the
        // first block lagged the scores -- adversarial inputs): the query tile again, reference following the maximum
        if (__ballot(bad)) {
            reset();
            general_softmax(first_blk);
            for (int blk = 1; blk < nblk; ++blk) {
                pv(blk - 1);
                general_softmax(blk);  // (O already holds block blk-1: the rescale applies to all of it)
            }
            pv(last_blk);
        }

        // ---- out[b, q, h*64 + d] = O^T[d][q] / l ; register v <-> channel (v&3) + 8*(v>>2) + 4*hf (+32)
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.0f / l_tot;
        // (token offsets inside one (batch, head) slice fit 32 bits -- the host checks it: scalar base + 32-bit lane offset)
        short *const obase = reinterpret_cast<short *>(a.out) + b * a.o_sb + h * a.o_sh + seg * a.o_seg;
        short *op = obase + ((unsigned)(qrow < a.N ? qrow : 0) * (unsigned)a.o_sn + 32u * hf);
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
            const auto s0w = __builtin_amdgcn_permlane32_swap(a2[0], b2[0], false, false);
            const auto s1w = __builtin_amdgcn_permlane32_swap(a2[1], b2[1], false, false);
            const uint4 row16 = uint4{s0w[0], s1w[0], s0w[1], s1w[1]};
            if (qrow < a.N) *reinterpret_cast<uint4 *>(op + 8 * g) = row16;
        }
        ATT_STAMP(5);  // this tile's stores issued
        if (qt + 8 < ntq) {  // the next tile's rows (wave-uniform); the other waves of the SIMD cover their latency
            const int qload = min((qt + 8) * 32 + col, a.N - 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qraw[ks] = *reinterpret_cast<const uint4 *>(qp + ((unsigned)qload * (unsigned)a.q_sn + 16u * ks + 8u * hf));
        }
    }
}
