// tome_attn.h -- proportional attention of the ToMe patches as one gfx950 kernel.
//
//   attn = softmax(q k^T * scale + log(size)[keys])  ;  out = attn v
//   (ToMeAttention.forward: tome/patch/videomae.py:55-66, vivit.py:95-113; timesformer.py:66-78 adds the bias to
//    the non-class block of the logits only; motionformer.py:98-121 takes one softmax per frame: `nseg` segments)
//
// PyTorch-ROCm's fused attention leaves its fast path as soon as a bias tensor is passed (measured on MI355X:
// 1037 us with a [B,1,1,N] bias vs 560 us without at 8 x 12 x 3137 x 64); here the per-key bias is one fp32
// value per key folded into the accumulators' start values, so proportional attention costs what plain attention
// costs plus 32 fma per tile.
//
// Structure (head dim 64, 16-bit q/k/v, fp32 softmax and accumulation):
//   * workgroup = 8 waves = 256 queries of one (batch, head, segment) (4 waves for short launches, two such
//     workgroups per CU); wave w owns queries 32w .. 32w+31; two waves per SIMD.
//   * keys/values stream through a two-slot LDS ring in tiles of 64 keys (register-staged: the loads of tile t+2
//     are issued in iteration t and written to LDS at the top of iteration t+1): ONE barrier per tile.
//   * S^T = K Q~^T on v_mfma_f32_32x32x16_bf16 (A = K rows from LDS, B = this wave's Q fragment in registers,
//     pre-multiplied by scale*log2(e)): accumulator register v of lane l holds key (v&3) + 8*(v>>2) + 4*(l>>5)
//     (+32 per key block) of query l&31, so a query's scores are lane-local (plus the partner lane l^32).
//   * the score accumulators START at -m_run (+ the key's bias): the matrix pipe delivers s - m_run and a weight is
//     one v_exp_f32 -- no scale, no subtraction, no running maximum in the steady state.  The weights are taken
//     against the reference point fixed by the first tile; O and l carry the same factor, so the result is exact as
//     long as nothing overflows, which the row sum (needed anyway) tells: a lane sum above AttLimit (or inf / NaN)
//     makes the workgroup repeat the block on the general path, whose reference follows the maximum tile by tile.
//   * software pipeline: iteration t multiplies S(t+1) and O += V(t) P(t) (16 MFMAs) while it turns S(t+1) into
//     P(t+1) (~80-110 vector instructions): the matrix work of an iteration does not depend on its vector work, so
//     the two interleave inside every wave instead of alternating in phases that all waves enter together.
//   * P^T (16-bit) is the B operand of O^T += V^T P^T as it sits in those registers; V^T fragments come from the
//     row-major V tile through ds_read_b64_tr_b16 (4 consecutive keys of one channel per lane).
//
// Measured on MI355X (bf16, random q/k/v; tools/attn_bench.py): plain 730 TFLOP/s at 128x12x1568, 810-830 at
// 128x12x1472, 890 at 8x12x3137 (round 1: 500-620); 12-layer VideoMAE-B mix of bench.py 776-790.  Against what the
// matrix pipe gives this dependency shape (tools/probes/mfma_bf16_probe.hip, same box): bare MFMA loop 1.60-1.73
// PFLOP/s, with the softmax's vector work per step (32 v_exp, 32 adds, 16 conversions feeding the next step)
// 1.14-1.22, a barrier per step costs nothing more.  Counters (profiles/r02_attention_pmc.json): matrix pipe busy
// 33 % at the 2.0 GHz the chip holds here, waves 35 % issuing / 27 % issue-stalled / 38 % parked.
// What was tried on this structure and makes no difference (within 1-2 %): a hand-placed issue order (every MFMA
// followed by the vector work that fits its shadow, sched_barrier fences), sched_group_barrier pipelines, persistent
// workgroups walking runs of items (-3 %), 5/6/7-wave workgroups to avoid a part-empty last block (-10...-30 %), the
// full 256-query blocks in one launch and the remaining queries in a second launch of 4-wave workgroups (-3...-12 %),
// 64 queries per wave -- four waves per workgroup, one per SIMD with ~500 registers, every K / V fragment serving two
// query blocks: 647 / 744 / 782 TFLOP/s where this form gives 721 / 818 / 878 (the compiler's schedule at that
// register count, not the idea, is what loses: the guide's 1.25 PF kernel of that shape is hand-placed assembly); reference
// point 0 for the whole first pass (no first-tile maximum, no start block: 708 / 771 / 825, short sequences unchanged).
// Ablations (results wrong by design, tools/ab_lib.sh): no v_exp +9 %, no barrier 0 %, no V-fragment reads +3 %,
// no staging at all +11 %; nothing but the MFMAs, row sums and conversions: +26 %.
// Pitfall met on the way: an `asm("v_add_f32 ...")` reading a v_exp_f32 result gets no hazard padding from hipcc
// (wrong sums on some launches); the additions are compiler instructions behind an empty asm barrier instead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tome_common.h"

#define ATT_D 64        // head dim
#define ATT_BN 64       // keys per tile
#define ATT_KS 72       // K tile row stride in elements (144 B: conflict-free ds_read_b128 over 16 rows)
#define ATT_VS 96       // V tile row stride in elements (192 B: conflict-free ds_read_b64_tr_b16 over 4 rows)

typedef float att_f32x16 __attribute__((ext_vector_type(16)));
typedef short att_s16x4 __attribute__((ext_vector_type(4)));
typedef short att_s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 att_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 att_f16x8 __attribute__((ext_vector_type(8)));

struct AttnArgs {
    const void *q, *k, *v;
    void *out;
    int64_t q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn;  // element strides: batch, head, token
    int64_t o_sb, o_sh, o_sn;                                      // out element strides (contiguous: [B, Nq, H*64])
    const float *log_size;                                         // NULL or [B, Nk - bias_skip] fp32
    int64_t ls_sb;
    int B, H, N, Nk;                                               // N queries, Nk keys per (batch, head)
    float scale;
    int bias_skip;  // 1: TimeSformer form -- key 0 and query 0 carry no bias, log_size[j-1] belongs to key j
    // Segmented form (Motionformer's per-frame attention, tome/patch/motionformer.py:98-121: every query attends
    // to the keys of ONE frame at a time, softmax per frame): nseg independent key ranges of Nk keys each, segment
    // s offset by s*k_seg / s*v_seg elements in k / v, s*ls_seg in log_size, and writing to out + s*o_seg.
    // nseg = 1: the plain form.  One launch instead of nseg.
    int nseg;
    int64_t k_seg, v_seg, o_seg, ls_seg;
};

template <typename TX> struct AttMfma;
template <> struct AttMfma<bf16_t> {
    static __device__ __forceinline__ att_f32x16 run(att_s16x8 a, att_s16x8 b, att_f32x16 c) {
        att_bf16x8 x, y;
        __builtin_memcpy(&x, &a, 16);
        __builtin_memcpy(&y, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
    }
};
template <> struct AttMfma<f16_t> {
    static __device__ __forceinline__ att_f32x16 run(att_s16x8 a, att_s16x8 b, att_f32x16 c) {
        att_f16x8 x, y;
        __builtin_memcpy(&x, &a, 16);
        __builtin_memcpy(&y, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
    }
};

// Largest lane sum of un-normalised weights the plain path accepts before it moves the reference point: the weights
// go to the second product in the 16-bit format (fp16 tops out at 65504; bf16 has fp32's range) and O / l
// accumulate N of them in fp32.
template <typename TX> struct AttLimit;
template <> struct AttLimit<bf16_t> { static constexpr float value = 1.152921504606847e18f; };  // 2^60
template <> struct AttLimit<f16_t> { static constexpr float value = 32768.0f; };                // 2^15

__device__ __forceinline__ float att_add(float x, float y) {
    // one v_add_f32 that the vectoriser cannot pair into v_pk_add_f32: the sum passes through an empty asm statement
    // (the add itself stays a compiler instruction, so its hazards -- a transcendental result read by the next
    // VALU instruction -- are padded by hipcc, which it would not do inside an asm string)
    float r = x + y;
    asm("" : "+v"(r));
    return r;
}

template <typename TX> __device__ __forceinline__ short att_bits(float f) {
    const TX t = from_f32<TX>(f);
    short s;
    __builtin_memcpy(&s, &t, 2);
    return s;
}

// ------------------------------------------------------------------------------------------------
// k_prop_attention<TX, WAVES, BIAS>: the kernel described at the top of this file.  BIAS = false: no per-key term
// (VideoMAE's default prop_attn = False; every patched attention before the first merge; Motionformer's class
// token).  BIAS = true: log(size) per key (a.log_size), TimeSformer's bias_skip form, Motionformer's segments.
//
//   iteration t:   S(t+1) = K(t+1) Q~^T + (bias - m_run)     8 MFMAs
//                  O     += V(t)^T P(t)^T                     8 MFMAs
//                  P(t+1) = exp2(S(t+1)), row sum, conversion to the 16-bit format
//
// First tile: general softmax (maximum, reference point).  Last two tiles (a full one left over by the unrolled
// pairs, a partly filled one): the same fast step with a run-time slot and out-of-range weights forced to zero.
// ------------------------------------------------------------------------------------------------
#define ATT_SLOTS 2
// s_setprio(1) while a wave runs the softmax of its step (exp, row sums, packing), 0 around its matrix instructions: the
// partner wave's MFMAs fill in behind.  Measured in round 2 (A/B libraries, one box): none 725-734 / 821-825 TFLOP/s plain,
// 651 / 690 with the size bias at 128x12x1568 / x1472; this form 736-737 / 830 and 665-667 / 710; tried and dropped:
// priority around the MFMAs instead 684 / 786; from the V-fragment reads on 726-728 / 781-792; exp only 721-723 /
// 788-791; a static priority for one wave of each SIMD pair: no gain plain, -5...-8 % with the bias.
// (The ablation switches of rounds 2-3 -- no v_exp, no barrier, no V-fragment reads, no staging -- are gone from this
// file; their results are in DESIGN_HISTORY.md.)

template <int V> struct AttInt { static constexpr int value = V; };

#ifdef ATT_DIAG
// Diagnostic build only (tools/attn_diag.py; never the shipped library): every wave of the first ATT_DIAG_WGS
// workgroups stamps the 100 MHz wall counter (s_memrealtime) at the phases of its query block; the stamps go to a
// buffer of their own that nothing else reads.
#define ATT_DIAG_WGS 8192
#define ATT_DIAG_N 8
__device__ unsigned long long g_att_stamps[ATT_DIAG_WGS * 8 * ATT_DIAG_N];
#define ATT_STAMP(i)                                                                                         \
    do {                                                                                                     \
        if (blockIdx.x < ATT_DIAG_WGS) {                                                                     \
            const unsigned long long st_ = __builtin_amdgcn_s_memrealtime();                                 \
            if (lane == 0) g_att_stamps[((size_t)blockIdx.x * 8 + wave) * ATT_DIAG_N + (i)] = st_;            \
        }                                                                                                    \
    } while (0)
#else
#define ATT_STAMP(i)
#endif

template <typename TX, int WAVES, bool BIAS>
__global__ __launch_bounds__(64 * WAVES, 2) void k_prop_attention(AttnArgs a) {  // two waves per SIMD either way
    constexpr int ATT_BM = 32 * WAVES;
    // two LDS slots: tile t lives in slot t & 1.  Tile t+1 is written during iteration t, when every wave has left
    // iteration t-2 -- the last one that read slot (t+1) & 1 (K(t-1) for its scores, V(t-1) into registers).
    __shared__ __attribute__((aligned(16))) short lds_k[ATT_SLOTS][ATT_BN * ATT_KS];
    __shared__ __attribute__((aligned(16))) short lds_v[ATT_SLOTS][ATT_BN * ATT_VS];
    __shared__ __attribute__((aligned(16))) float lds_bias[ATT_SLOTS][ATT_BN];  // BIAS: log(size) * log2(e) per key

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, hf = lane >> 5;
    ATT_STAMP(0);  // entry
#ifdef ATT_DIAG
    if (blockIdx.x < ATT_DIAG_WGS && lane == 0)
        g_att_stamps[((size_t)blockIdx.x * 8 + wave) * ATT_DIAG_N + 7] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
#endif
    const int qblocks = (a.N + ATT_BM - 1) / ATT_BM;
    // XCD-aware mapping: the query blocks of one (batch, head, segment) stream the same K/V -> same XCD (ids congruent
    // mod 8).  (Measured: persistent workgroups walking a run of items each are 3 % slower than this.)
    const int L = blockIdx.x;
    const int xcd = L & 7, sq = L >> 3;
    const int bhs = (sq / qblocks) * 8 + xcd;
    if (bhs >= a.B * a.H * a.nseg) return;
    const int qb = sq % qblocks;
    const int bh = bhs / a.nseg, seg = bhs - bh * a.nseg;
    const int b = bh / a.H, h = bh % a.H;
    const short *qp = reinterpret_cast<const short *>(a.q) + b * a.q_sb + h * a.q_sh;
    const short *kp = reinterpret_cast<const short *>(a.k) + b * a.k_sb + h * a.k_sh + seg * a.k_seg;
    const short *vp = reinterpret_cast<const short *>(a.v) + b * a.v_sb + h * a.v_sh + seg * a.v_seg;
    const float *lsp = BIAS ? a.log_size + b * a.ls_sb + seg * a.ls_seg : nullptr;

    const int qrow = qb * ATT_BM + wave * 32 + col;
    const int qload = qrow < a.N ? qrow : a.N - 1;
    const bool active = qb * ATT_BM + __builtin_amdgcn_readfirstlane(wave) * 32 < a.N;  // wave-uniform (scalar)
    const float LOG2E = 1.4426950408889634f;
    // TimeSformer form (bias_skip): key 0 and query 0 carry no bias, log_size[j-1] belongs to key j -- the class query's
    // lane multiplies every key's bias by 0
    const float bfac = (a.bias_skip && qrow == 0) ? 0.0f : 1.0f;
    const float sl = a.scale * LOG2E;
    att_s16x8 qf[4];  // q * scale * log2(e), rounded once to the 16-bit format (see the top of this file)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const att_s16x8 raw = *reinterpret_cast<const att_s16x8 *>(qp + (int64_t)qload * a.q_sn + 16 * ks + 8 * hf);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            TX tq;
            const short r = raw[e];
            __builtin_memcpy(&tq, &r, 2);
            qf[ks][e] = att_bits<TX>(to_f32(tq) * sl);
        }
    }

    ATT_STAMP(1);  // Q fragment requested / in registers
    att_f32x16 o0, o1, negm;
#pragma unroll
    for (int v = 0; v < 16; ++v) o0[v] = o1[v] = negm[v] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const int ntiles = (a.Nk + ATT_BN - 1) / ATT_BN;
    const int nfull = a.Nk / ATT_BN;  // tiles 0 .. nfull-1 hold 64 keys

    // staging through registers: thread -> rows r0 (+ RSTEP), 16-byte column c0 of the 64 x 64 K and V tiles
    constexpr int RSTEP = 8 * WAVES;
    constexpr int NST = ATT_BN / RSTEP;  // WAVES is 4 or 8
    const int r0 = tid >> 3, c0 = tid & 7;
    uint4 kreg[NST], vreg[NST];
    float breg = 0.0f;
    auto stage_load = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int key = t * ATT_BN + r0 + RSTEP * i;
            if (key < a.Nk) {
                kreg[i] = *reinterpret_cast<const uint4 *>(kp + (int64_t)key * a.k_sn + 8 * c0);
                vreg[i] = *reinterpret_cast<const uint4 *>(vp + (int64_t)key * a.v_sn + 8 * c0);
            } else {
                kreg[i] = uint4{0, 0, 0, 0};
                vreg[i] = uint4{0, 0, 0, 0};  // zeros: weight 0 times a finite value
            }
        }
        if (BIAS && tid < ATT_BN) {
            const int key = t * ATT_BN + tid;
            breg = (key < a.Nk && key >= a.bias_skip) ? lsp[key - a.bias_skip] * LOG2E : 0.0f;
        }
    };
    // (the lambdas below take the LDS slot as a plain int: inlined with a literal it folds into the instructions'
    // immediate offsets -- the fast loop; with a run-time value it costs an address add -- the general path)
    short *const wk = &lds_k[0][0] + r0 * ATT_KS + 8 * c0, *const wv = &lds_v[0][0] + r0 * ATT_VS + 8 * c0;
    auto stage_write = [&](int S) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            *reinterpret_cast<uint4 *>(wk + S * ATT_BN * ATT_KS + RSTEP * i * ATT_KS) = kreg[i];
            *reinterpret_cast<uint4 *>(wv + S * ATT_BN * ATT_VS + RSTEP * i * ATT_VS) = vreg[i];
        }
        if (BIAS && tid < ATT_BN) lds_bias[0][S * ATT_BN + tid] = breg;
    };
    // V^T fragments of a whole tile (layout: see O += V P below); lane-constant base
    typedef __attribute__((address_space(3))) att_s16x4 *lds_s16x4_p;
    const short *const vbase = &lds_v[0][0] + (4 * hf + ((lane & 15) >> 2)) * ATT_VS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    att_s16x4 vfr[2][2][4];
    auto v_fragments = [&](int S) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const short *va = vbase + S * ATT_BN * ATT_VS + (32 * kb + 16 * p) * ATT_VS;
                vfr[kb][p][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va));
                vfr[kb][p][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 8 * ATT_VS));
                vfr[kb][p][2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 32));
                vfr[kb][p][3] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 8 * ATT_VS + 32));
            }
    };
    const short *const kbase = &lds_k[0][0] + col * ATT_KS + 8 * hf;
    att_f32x16 s0, s1;
    // per-key bias of this lane's registers (register v <-> key (v&3) + 8*(v>>2) + 4*hf, +32 for the second block)
    // added to a start block: the score accumulators then deliver s + log2e*log(size) - m_run
    auto add_bias = [&](int S, att_f32x16 &c0, att_f32x16 &c1) __attribute__((always_inline)) {
        const float *brow = &lds_bias[0][0] + S * ATT_BN + 4 * hf;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b0 = *reinterpret_cast<const float4 *>(brow + 8 * g);
            const float4 b1 = *reinterpret_cast<const float4 *>(brow + 32 + 8 * g);
            c0[4 * g + 0] = __builtin_fmaf(bfac, b0.x, c0[4 * g + 0]); c0[4 * g + 1] = __builtin_fmaf(bfac, b0.y, c0[4 * g + 1]);
            c0[4 * g + 2] = __builtin_fmaf(bfac, b0.z, c0[4 * g + 2]); c0[4 * g + 3] = __builtin_fmaf(bfac, b0.w, c0[4 * g + 3]);
            c1[4 * g + 0] = __builtin_fmaf(bfac, b1.x, c1[4 * g + 0]); c1[4 * g + 1] = __builtin_fmaf(bfac, b1.y, c1[4 * g + 1]);
            c1[4 * g + 2] = __builtin_fmaf(bfac, b1.z, c1[4 * g + 2]); c1[4 * g + 3] = __builtin_fmaf(bfac, b1.w, c1[4 * g + 3]);
        }
    };
    // (first_half_only: keys 32..63 of the tile lie past the end -- their weights are forced to zero anyway, so their
    // four matrix instructions are left out; wave-uniform)
    auto scores = [&](int S, const att_f32x16 &cinit, bool first_half_only = false) __attribute__((always_inline)) {
        // block 0 (keys 0..31) completes before block 1 starts: its weights can be taken while block 1 multiplies
        s0 = cinit;
        s1 = cinit;
        if (BIAS) add_bias(S, s0, s1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            s0 = AttMfma<TX>::run(*reinterpret_cast<const att_s16x8 *>(kbase + S * ATT_BN * ATT_KS + 16 * ks), qf[ks], s0);
        if (!first_half_only) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s1 = AttMfma<TX>::run(*reinterpret_cast<const att_s16x8 *>(kbase + S * ATT_BN * ATT_KS + 32 * ATT_KS + 16 * ks),
                                      qf[ks], s1);
        }
    };
    // P (fp32 weights in s0, s1) -> the four 16-bit B fragments of O^T += V^T P^T
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
        scores(S, zero);
        const int key0 = t * ATT_BN + 4 * hf;
        float mt = -INFINITY;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int key = key0 + (v & 3) + 8 * (v >> 2);
            s0[v] = key < a.Nk ? s0[v] : -INFINITY;
            s1[v] = key + 32 < a.Nk ? s1[v] : -INFINITY;
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
    // O^T += V(t)^T P(t)^T from the fragments in registers (first_half_only: the weights of keys 32..63 are zero)
    auto pv = [&](bool first_half_only = false) __attribute__((always_inline)) {
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
    // One iteration = tile t (weights and V fragments in registers) is accumulated while tile t+1 (slot S after the
    // barrier) gets its scores and weights.
    //   fast_step: tile t+1 holds 64 keys.  Straight-line code -- 16 MFMAs and ~80 vector instructions free to
    //   interleave.  The overflow guard only sets a flag (see below).
    bool bad = false;
    auto fast_step = [&](int S, int t, bool masked) __attribute__((always_inline)) {
        stage_write(S);  // registers hold tile t+1
        // tile t+2, rows clamped to the last key instead of a bounds branch (a partly filled tile gets its scores
        // masked by the general softmax; its weights are 0, so a repeated V row adds nothing): no control flow
        // between two barriers
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int key = min((t + 2) * ATT_BN + r0 + RSTEP * i, a.Nk - 1);
            kreg[i] = *reinterpret_cast<const uint4 *>(kp + (int64_t)key * a.k_sn + 8 * c0);
            vreg[i] = *reinterpret_cast<const uint4 *>(vp + (int64_t)key * a.v_sn + 8 * c0);
        }
        if (BIAS && tid < ATT_BN) {
            const int key = min((t + 2) * ATT_BN + tid, a.Nk - 1);
            breg = key >= a.bias_skip ? lsp[key - a.bias_skip] * LOG2E : 0.0f;
        }
        __syncthreads();   // tile t+1 visible; every wave has left iteration t-1
        // (measured: a hand-placed issue order of this block -- every MFMA followed by the vector work that fits its
        // shadow, pinned by sched_barrier fences -- runs within 1 % of what the compiler makes of it)
        // the partly filled last tile with at most 32 keys in range: the second half of its scores is never computed
        const bool half = masked && (t + 1) * ATT_BN + 32 >= a.Nk;
        scores(S, negm, half);
        pv();
        v_fragments(S);
        if (!(BIAS && WAVES == 4)) __builtin_amdgcn_s_setprio(1);  // (that instance would spill: the builtin fences the scheduler)
#pragma unroll
        for (int v = 0; v < 16; ++v) s0[v] = __builtin_amdgcn_exp2f(s0[v]);
#pragma unroll
        for (int v = 0; v < 16; ++v) s1[v] = __builtin_amdgcn_exp2f(s1[v]);
        if (masked) {  // the partly filled last tile: keys past the end weigh nothing
            const int key0 = (t + 1) * ATT_BN + 4 * hf;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int key = key0 + (v & 3) + 8 * (v >> 2);
                s0[v] = key < a.Nk ? s0[v] : 0.0f;
                s1[v] = key + 32 < a.Nk ? s1[v] : 0.0f;
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
        if (!(BIAS && WAVES == 4)) __builtin_amdgcn_s_setprio(0);
        l_run += lsum;
        bad = bad || !(lsum <= AttLimit<TX>::value);  // inf / NaN / too large: this pass is void (rerun below)
    };
    //   slow_step: any tile t+1 (partly filled, or the general pass): general softmax, run-time slot.
    auto slow_step = [&](int t) __attribute__((always_inline)) {
        const int S = (t + 1) & 1;
        stage_write(S);
        if (t + 2 < ntiles) stage_load(t + 2);
        __syncthreads();
        pv();
        v_fragments(S);
        general_softmax(t + 1, S);  // (O already holds tile t: the rescale applies to all of it)
    };
    //   a wave that owns no query only stages and meets the barriers
    auto helper_step = [&](int t) __attribute__((always_inline)) {
        stage_write((t + 1) & 1);
        if (t + 2 < ntiles) stage_load(t + 2);
        __syncthreads();
    };

    // Pass 0 takes the fast form wherever a tile is full.  Should a row sum have tripped the guard in any wave of
    // the workgroup (weights beyond the 16-bit format's range or an overflow: the running reference point lagged
    // the scores by more than the format allows -- adversarial inputs), the workgroup repeats the block on the
    // general path (pass 1), whose reference point follows the maximum tile by tile.
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            if (!__syncthreads_or(bad ? 1 : 0)) break;
#pragma unroll
            for (int v = 0; v < 16; ++v) o0[v] = o1[v] = negm[v] = 0.0f;
            m_run = -INFINITY;
            l_run = 0.0f;
        }
        // ---- prologue: tile 0 in slot 0, tile 1 on its way
        stage_load(0);
        stage_write(0);
        if (ntiles > 1) stage_load(1);
        __syncthreads();
        ATT_STAMP(2);  // tile 0 in LDS
        // iterations t = 0 .. ntiles-2 bring in tile t+1; those with t+1 < nfull may take the fast form
        int t = 0;
        if (active) {
            general_softmax(0, 0);
            v_fragments(0);
            ATT_STAMP(3);  // first tile's weights
            if (pass == 0) {
                for (; t + 2 < nfull; t += 2) {  // t even: tile t+1 -> slot 1, tile t+2 -> slot 0, both full
                    fast_step(1, t, false);
                    fast_step(0, t + 1, false);
                }
                // at most two tiles left -- a last full one, a partly filled one: the same step with a run-time slot
                // and the weights of keys past the end forced to zero (one more copy of the code, not four)
                for (; t + 1 < ntiles; ++t) fast_step((t + 1) & 1, t, true);
            }
            for (; t + 1 < ntiles; ++t) slow_step(t);
            ATT_STAMP(4);  // tile loop done
            pv((ntiles - 1) * ATT_BN + 32 >= a.Nk);  // the last tile (its keys 32..63 may all lie past the end)
        } else {
            for (; t + 1 < ntiles; ++t) helper_step(t);
        }
    }

    // ---- out[b, q, h*64 + d] = O^T[d][q] / l ; register v <-> channel (v&3) + 8*(v>>2) + 4*hf (+32)
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    // A lane holds channels 8g + 4*hf .. +3 (w0) and 32 + 8g + 4*hf .. +3 (w1) of its query; its partner lane l^32 holds
    // the other halves.  One v_permlane32_swap per dword hands the lower lane the partner's w0 and the upper lane the
    // partner's w1, so every lane stores 16 contiguous bytes (channels 8g .. 8g+7, +32 for the upper lanes): four
    // 16-byte stores per lane instead of eight 8-byte ones.  (All 64 lanes take part in the swaps; only the stores are
    // predicated.)
    short *op = reinterpret_cast<short *>(a.out) + b * a.o_sb + (int64_t)(qrow < a.N ? qrow : 0) * a.o_sn + h * a.o_sh +
                seg * a.o_seg + 32 * hf;
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
    ATT_STAMP(5);  // stores issued
#ifdef ATT_DIAG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_STAMP(6);  // stores done
#endif
}

// ------------------------------------------------------------------------------------------------
// k_trajectory_mix: the temporal stage of Motionformer's trajectory attention as one streaming pass
// (ToMeTrajectoryAttention.forward, tome/patch/motionformer.py:122-139):
//     tattn = softmax_f( (q2 * scale) . k2[f] )          one logit per frame of the token's trajectory
//     out   = sum_f tattn[f] * val[f]
// per (batch, token, head).  q2 [B, S, H*64]; k2 and val [B, S, F, H*64] views with a row stride each (k2 is the
// first half of the proj_kv output, val the trajectory tokens y or the second half); out [B, S, H*64];
// tattn [B, H, S, F] fp32 (optional).  One wave per token: lane l owns the 16-byte chunks l and l + 64 of the
// H*64 channels (H <= 16), the 8 lanes of a head reduce their partial dot products by xor-shuffles, the F <= 16
// weights of a head live in registers, everything is loaded before it is used.  HBM bound: k2 and val are each
// read once -- the five element-wise / reduction passes it replaces moved them three times.
// ------------------------------------------------------------------------------------------------
#define TRAJ_MAXF 8
#ifndef TRAJ_NT
#define TRAJ_NT 1  // measured at 64 x 1568 x 8 x 768: 522 -> 511 us (5.31 -> 5.43 TB/s)
#endif
__device__ __forceinline__ uint4 traj_ld16(const void *p) {  // k2 / val are read exactly once
    if (TRAJ_NT) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        return uint4{v.x, v.y, v.z, v.w};
    }
    return *reinterpret_cast<const uint4 *>(p);
}

template <typename TX>
__global__ __launch_bounds__(256) void k_trajectory_mix(const TX *__restrict__ q2, const TX *__restrict__ k2,
                                                        const TX *__restrict__ val, int64_t rows, int S, int F, int H,
                                                        int64_t k_row, int64_t v_row, float scale,
                                                        TX *__restrict__ out, int64_t out_sb, float *__restrict__ tattn) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= rows) return;  // wave-uniform
    const int C = H * 64, chunks = C >> 3;
    const TX *qr = q2 + row * C;
    const TX *kr = k2 + row * F * k_row;
    const TX *vr = val + row * F * v_row;
    // (out rows: batch b starts out_sb elements after batch b-1 -- the caller may hand a slice of a larger buffer)
    const int64_t ob = row / S;
    TX *const orow = out + ob * out_sb + (row - ob * S) * C;
    float acc[2][8];
    float w[2][TRAJ_MAXF];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = lane + 64 * i;
        const bool on = c < chunks;
        // Every load is unconditional (a lane without a chunk reads chunk 0, a frame past F reads frame F-1, both
        // ignored): with the loads inside `if (on && f < F)` hipcc put an s_waitcnt vmcnt(0) in front of every load of
        // the second group, one round trip per frame.  k2 AND val are requested before any arithmetic.
        const int cl = on ? c : 0;
        uint4 kraw[TRAJ_MAXF], vraw[TRAJ_MAXF];
#pragma unroll
        for (int f = 0; f < TRAJ_MAXF; ++f) kraw[f] = traj_ld16(kr + (int64_t)(f < F ? f : F - 1) * k_row + 8 * cl);
        const uint4 qraw = *reinterpret_cast<const uint4 *>(qr + 8 * cl);
#pragma unroll
        for (int f = 0; f < TRAJ_MAXF; ++f) vraw[f] = traj_ld16(vr + (int64_t)(f < F ? f : F - 1) * v_row + 8 * cl);
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler would otherwise sink the val loads below the dot products)
        float qv[8];
        {
            Pack<TX, 8> pq;
            __builtin_memcpy(&pq, &qraw, 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[e] = to_f32(pq.e[e]);
        }
        float lg[TRAJ_MAXF];
#pragma unroll
        for (int f = 0; f < TRAJ_MAXF; ++f) {
            float d = 0.0f;
            {
                Pack<TX, 8> pk;
                __builtin_memcpy(&pk, &kraw[f], 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) d = __builtin_fmaf(qv[e], to_f32(pk.e[e]), d);
            }
            // the 8 lanes of a head (consecutive chunks) hold its 64 channels
            d += __shfl_xor(d, 1);
            d += __shfl_xor(d, 2);
            d += __shfl_xor(d, 4);
            lg[f] = (f < F) ? d * scale : -INFINITY;
        }
        float m = lg[0];
#pragma unroll
        for (int f = 1; f < TRAJ_MAXF; ++f) m = fmaxf(m, lg[f]);
        float sum = 0.0f;
#pragma unroll
        for (int f = 0; f < TRAJ_MAXF; ++f) {
            w[i][f] = (f < F) ? __expf(lg[f] - m) : 0.0f;
            sum += w[i][f];
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int f = 0; f < TRAJ_MAXF; ++f) w[i][f] *= inv;
        if (tattn && on && (c & 7) == 0) {
            const int h = c >> 3;
            float *tp = tattn + ((ob * H + h) * S + (row - ob * S)) * F;
            for (int f = 0; f < F; ++f) tp[f] = w[i][f];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = 0.0f;
#pragma unroll
        for (int f = 0; f < TRAJ_MAXF; ++f) {
            if (f < F) {  // (wave-uniform)
                Pack<TX, 8> pk;
                __builtin_memcpy(&pk, &vraw[f], 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][e] = __builtin_fmaf(w[i][f], to_f32(pk.e[e]), acc[i][e]);
            }
        }
        if (on) store_pack<TX, 8>(orow + 8 * c, acc[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// k_short_attention: attention over sequences of at most 8 tokens -- TimeSformer's temporal attention, the T copies of
// one spatial token (ToMeBlock.forward, tome/patch/timesformer.py:25-27: self.temporal_attn on 'b (p t) m ->
// (b p) t m'; the module is the host model's softmax(q k^T * scale) v).  Thousands of 8 x 8 problems per launch: no
// matrix pipe, HBM bound (q, k, v read once from the qkv buffer in place, out written once as [B, N, H*64]).
// Eight lanes own one (sequence, head): lane c holds channels 8c .. 8c+7 of every token's q, k and v row (3 x N
// 16-byte loads, all issued before use); the N x N logits are packed 2-element dot products (v_dot2: exact products,
// fp32 sums) reduced over the eight lanes by xor-shuffles; softmax and the weighted sum of v in fp32.
// ------------------------------------------------------------------------------------------------
#define SHORT_MAXN 8
typedef __bf16 short_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 short_f16x2 __attribute__((ext_vector_type(2)));
template <typename TX> __device__ __forceinline__ float short_dot8(const uint4 &x, const uint4 &y);
template <> __device__ __forceinline__ float short_dot8<bf16_t>(const uint4 &x, const uint4 &y) {
    const uint32_t a[4] = {x.x, x.y, x.z, x.w}, b[4] = {y.x, y.y, y.z, y.w};
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        short_bf16x2 p, q;
        __builtin_memcpy(&p, &a[i], 4);
        __builtin_memcpy(&q, &b[i], 4);
        t = __builtin_amdgcn_fdot2_f32_bf16(p, q, t, false);
    }
    return t;
}
template <> __device__ __forceinline__ float short_dot8<f16_t>(const uint4 &x, const uint4 &y) {
    const uint32_t a[4] = {x.x, x.y, x.z, x.w}, b[4] = {y.x, y.y, y.z, y.w};
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        short_f16x2 p, q;
        __builtin_memcpy(&p, &a[i], 4);
        __builtin_memcpy(&q, &b[i], 4);
        t = __builtin_amdgcn_fdot2(p, q, t, false);
    }
    return t;
}

template <typename TX>
__global__ __launch_bounds__(256) void k_short_attention(const TX *__restrict__ q, const TX *__restrict__ k,
                                                         const TX *__restrict__ v, int64_t q_sb, int64_t q_sn,
                                                         int64_t k_sb, int64_t k_sn, int64_t v_sb, int64_t v_sn,
                                                         int64_t units, int H, int N, float scale,
                                                         TX *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t unit = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    const bool on = unit < units;
    const int64_t u = on ? unit : units - 1;  // (a lane past the end repeats the last unit's loads, stores nothing)
    const int64_t b = u / H;
    const int h = (int)(u - b * H);
    const int ch = h * 64 + 8 * (lane & 7);
    const TX *qr = q + b * q_sb + ch, *kr = k + b * k_sb + ch, *vr = v + b * v_sb + ch;
    uint4 qraw[SHORT_MAXN], kraw[SHORT_MAXN], vraw[SHORT_MAXN];
#pragma unroll
    for (int t = 0; t < SHORT_MAXN; ++t) {
        const int tt = t < N ? t : N - 1;  // (wave-uniform; a token past the end repeats the last one, masked below)
        qraw[t] = traj_ld16(qr + tt * q_sn);
        kraw[t] = traj_ld16(kr + tt * k_sn);
    }
#pragma unroll
    for (int t = 0; t < SHORT_MAXN; ++t) vraw[t] = traj_ld16(vr + (t < N ? t : N - 1) * v_sn);
    const float LOG2E = 1.4426950408889634f;
    const float sl = scale * LOG2E;
    TX *orow = out + (b * N) * ((int64_t)H * 64) + ch;
#pragma unroll
    for (int i = 0; i < SHORT_MAXN; ++i) {
        if (i >= N) break;  // (wave-uniform)
        float d[SHORT_MAXN];
#pragma unroll
        for (int j = 0; j < SHORT_MAXN; ++j) {
            float t = short_dot8<TX>(qraw[i], kraw[j]);
            t += __shfl_xor(t, 1);
            t += __shfl_xor(t, 2);
            t += __shfl_xor(t, 4);
            d[j] = j < N ? t * sl : -INFINITY;
        }
        float m = d[0];
#pragma unroll
        for (int j = 1; j < SHORT_MAXN; ++j) m = fmaxf(m, d[j]);
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < SHORT_MAXN; ++j) {
            d[j] = __builtin_amdgcn_exp2f(d[j] - m);
            sum += d[j];
        }
        const float inv = 1.0f / sum;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int j = 0; j < SHORT_MAXN; ++j) {
            Pack<TX, 8> pv;
            __builtin_memcpy(&pv, &vraw[j], 16);
            const float w = d[j] * inv;  // (0 for a masked key)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(w, to_f32(pv.e[e]), acc[e]);
        }
        if (on) store_pack<TX, 8>(orow + (int64_t)i * H * 64, acc);
    }
}
