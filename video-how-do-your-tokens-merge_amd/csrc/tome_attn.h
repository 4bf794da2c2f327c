// tome_attn.h -- proportional attention of the ToMe patches as one gfx950 kernel.
//
//   attn = softmax(q k^T * scale + log(size)[keys])  ;  out = attn v
//   (ToMeAttention.forward: tome/patch/videomae.py:55-66, vivit.py:95-113; timesformer.py:66-78 adds the bias to
//    the non-class block of the logits only)
//
// PyTorch-ROCm's fused attention leaves its fast path as soon as a bias tensor is passed (measured on MI355X:
// 1037 us with a [B,1,1,N] bias vs 560 us without at 8 x 12 x 3137 x 64); here the per-key bias is one fp32
// value per key added in the softmax, so proportional attention costs what plain attention costs.
//
// Structure (head dim 64, 16-bit q/k/v, fp32 softmax and accumulation):
//   * workgroup = 8 waves = 256 queries of one (batch, head) (4 waves for sequences up to 128); wave w owns
//     queries 32w .. 32w+31.
//   * keys/values stream through LDS in tiles of 64 keys (register-staged: the next tile's global loads are
//     issued before the current tile's math and written to LDS behind it).
//   * S^T = K Q^T on v_mfma_f32_32x32x16_bf16 (A = K rows from LDS, B = this wave's Q fragment in registers):
//     accumulator register v of lane l holds key (v&3) + 8*(v>>2) + 4*(l>>5) (+32 per key block) of query l&31,
//     so a query's scores are lane-local (plus the partner lane l^32): max / sum / rescale need no LDS.
//   * P^T (bf16) is the B operand of O^T += V^T P^T as it sits in those registers; V^T fragments come from the
//     row-major V tile through ds_read_b64_tr_b16 (4 consecutive keys of one channel per lane).
//   * O^T accumulator: all 32 registers of a lane belong to its query -> the online-softmax rescale is a
//     lane-local multiply.
// Measured on MI355X (bf16): 500-570 TFLOP/s (8x12x3137: 450-470 us, 64x12x1568: 850-900 us) against 415-500 for
// the framework's kernel without a bias and 230-290 with one.  What was tried and makes no difference (within
// 1-3 %): one instead of two barriers per tile (double-buffered LDS), two vs four waves per SIMD, a deferred
// rescale of O, the scores of tile t+1 computed ahead of the softmax of tile t (for all waves, or only for the second wave of
// every SIMD so that the two start each interval in different pipes), static wave priorities, sixteen waves per
// workgroup.
// What does: for plain attention the weights are taken against the current reference point before the tile's
// maximum is known (one fma + one exp per score, no subtraction, no rescale while the maximum grows by less than
// 2^ATT_DEFER; +2 %); eight waves (256 queries) per workgroup sharing each staged tile (+4-9 %) -- staging K/V through
// registers into LDS costs 18 % of the time with four.  Where the rest goes (ablations on the eight-wave form): the
// 33 v_exp_f32 per tile 17 %, the 32 subtractions 5 %, the 32 row-sum additions 4 %: per 64-key tile a wave
// issues ~165 VALU + 33 transcendental + 16 conversion + 36 LDS instructions next to its 16 MFMAs (~1300 issue
// cycles against ~1700 measured per SIMD), i.e. the softmax arithmetic, not the matrix pipe (22 % busy), bounds it.
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

template <typename TX, int WAVES, bool BIAS = true>
__global__ __launch_bounds__(64 * WAVES) void k_prop_attention(AttnArgs a) {
    constexpr int ATT_BM = 32 * WAVES;  // queries per workgroup
    __shared__ __attribute__((aligned(16))) short lds_k[ATT_BN * ATT_KS];
    __shared__ __attribute__((aligned(16))) short lds_v[ATT_BN * ATT_VS];
    __shared__ __attribute__((aligned(16))) float lds_bias[ATT_BN];  // log(size)*log2(e) per key, -inf out of range
    __shared__ __attribute__((aligned(16))) float lds_mask[ATT_BN];  // 0 in range, -inf out of range

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, hf = lane >> 5;
    const int qblocks = (a.N + ATT_BM - 1) / ATT_BM;
    // XCD-aware mapping: the query blocks of one (batch, head) stream the same K/V -> same XCD (ids congruent mod 8)
    const int L = blockIdx.x;
    const int xcd = L & 7, s = L >> 3;
    const int bhs = (s / qblocks) * 8 + xcd;  // (batch, head, segment): one K/V stream each
    if (bhs >= a.B * a.H * a.nseg) return;
    const int qb = s % qblocks;
    const int bh = bhs / a.nseg, seg = bhs - bh * a.nseg;
    const int b = bh / a.H, h = bh % a.H;

    const short *qp = reinterpret_cast<const short *>(a.q) + b * a.q_sb + h * a.q_sh;
    const short *kp = reinterpret_cast<const short *>(a.k) + b * a.k_sb + h * a.k_sh + seg * a.k_seg;
    const short *vp = reinterpret_cast<const short *>(a.v) + b * a.v_sb + h * a.v_sh + seg * a.v_seg;
    const float *lsp = a.log_size ? a.log_size + b * a.ls_sb + seg * a.ls_seg : nullptr;

    // this lane's query and its Q fragment: B operand of S^T = K Q^T (k = channel): 4 steps x 8 channels
    const int qrow = qb * ATT_BM + wave * 32 + col;
    const int qload = qrow < a.N ? qrow : a.N - 1;
    // The fragment holds q * scale * log2(e) rounded ONCE to the 16-bit format (the reference rounds `q * self.scale`
    // to it as well, tome/patch/videomae.py:58): the MFMA then delivers the logits in base-2 units and the softmax
    // needs no multiply per score.
    const float LOG2E = 1.4426950408889634f;
    const float sl = a.scale * LOG2E;
    att_s16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const att_s16x8 raw = *reinterpret_cast<const att_s16x8 *>(qp + (int64_t)qload * a.q_sn + 16 * ks + 8 * hf);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            TX t;
            const short r = raw[e];
            __builtin_memcpy(&t, &r, 2);
            qf[ks][e] = att_bits<TX>(to_f32(t) * sl);
        }
    }
    const bool active = qb * ATT_BM + wave * 32 < a.N;  // wave-uniform: does this wave own any query at all?
    const bool unbiased_query = a.bias_skip && qrow == 0;
    const float *bias_row = unbiased_query ? lds_mask : lds_bias;
    att_f32x16 o0, o1;
#pragma unroll
    for (int v = 0; v < 16; ++v) o0[v] = o1[v] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    att_f32x16 negm;  // plain attention: -m_run in every register, the C operand the score MFMAs start from
#pragma unroll
    for (int v = 0; v < 16; ++v) negm[v] = 0.0f;

    const int ntiles = (a.Nk + ATT_BN - 1) / ATT_BN;
    // staging: thread t moves chunks c = t and t + 256 (16 B each) of the 64 x 64 K and V tiles
    constexpr int RSTEP = 8 * WAVES;         // rows covered by one pass of the workgroup
    constexpr int NST = (ATT_BN + RSTEP - 1) / RSTEP;  // passes: 16-byte chunks of each tile per thread (1 with 8 waves)
    constexpr bool EVEN = (ATT_BN % RSTEP) == 0;       // 5, 6, 7 waves: the last pass covers fewer rows
    const int r0 = tid >> 3, c0 = tid & 7;   // rows r0 (+ RSTEP), 16-byte column c0
    uint4 kreg[NST], vreg[NST];
    float breg = 0.0f, mreg = 0.0f;
    auto stage_load = [&](int t) {
        const int key0 = t * ATT_BN;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int key = key0 + r0 + RSTEP * i;
            if (!EVEN && r0 + RSTEP * i >= ATT_BN) continue;
            if (key < a.Nk) {
                kreg[i] = *reinterpret_cast<const uint4 *>(kp + (int64_t)key * a.k_sn + 8 * c0);
                vreg[i] = *reinterpret_cast<const uint4 *>(vp + (int64_t)key * a.v_sn + 8 * c0);
            } else {
                kreg[i] = uint4{0, 0, 0, 0};
                vreg[i] = uint4{0, 0, 0, 0};  // zeros: p = 0 times a finite value
            }
        }
        if (tid < ATT_BN) {
            const int key = key0 + tid;
            const bool in = key < a.Nk;
            float bv = 0.0f;
            if (in && lsp && key >= a.bias_skip) bv = lsp[key - a.bias_skip] * LOG2E;
            breg = in ? bv : -INFINITY;
            mreg = in ? 0.0f : -INFINITY;
        }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            if (!EVEN && r0 + RSTEP * i >= ATT_BN) continue;
            *reinterpret_cast<uint4 *>(lds_k + (r0 + RSTEP * i) * ATT_KS + 8 * c0) = kreg[i];
            *reinterpret_cast<uint4 *>(lds_v + (r0 + RSTEP * i) * ATT_VS + 8 * c0) = vreg[i];
        }
        if (tid < ATT_BN) {
            lds_bias[tid] = breg;
            lds_mask[tid] = mreg;
        }
    };

    stage_load(0);
    stage_write();
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) stage_load(t + 1);

        // ---- V^T fragments of the whole tile are requested first so that their LDS latency passes under the scores
        //      and the softmax: channel row = col (+32), one transposed read delivers 4 consecutive keys of one
        //      channel: lane i = 4*rq + pc of a 16-lane group addresses row rq, columns 4*pc .. 4*pc+3 of a 4 x 16
        //      block and receives column i.  Step (kb, p) contracts key slots {8*hf' + e} = keys 32*kb + 16*p +
        //      8*(e>>2) + 4*hf' + (e&3): the keys a lane holds in registers 8p .. 8p+7 of its score block kb.
        att_s16x4 vfr[2][2][4];
        auto v_fragments = [&]() {
            typedef __attribute__((address_space(3))) att_s16x4 *lds_s16x4_p;
            const int grp = (lane >> 4) & 1, li = lane & 15, rq = li >> 2, pc = li & 3;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const short *va = lds_v + (32 * kb + 16 * p + 4 * hf + rq) * ATT_VS + 16 * grp + 4 * pc;
                    vfr[kb][p][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va));
                    vfr[kb][p][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 8 * ATT_VS));
                    vfr[kb][p][2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 32));
                    vfr[kb][p][3] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 8 * ATT_VS + 32));
                }
        };
        // ---- S^T = K Q~^T (+ c0): two blocks of 32 keys, four channel steps; logits in base-2 units
        att_f32x16 s0, s1;
        auto scores = [&](const att_f32x16 &c0) {
            s0 = c0;
            s1 = c0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const att_s16x8 k0 = *reinterpret_cast<const att_s16x8 *>(lds_k + col * ATT_KS + 16 * ks + 8 * hf);
                const att_s16x8 k1 = *reinterpret_cast<const att_s16x8 *>(lds_k + (32 + col) * ATT_KS + 16 * ks + 8 * hf);
                s0 = AttMfma<TX>::run(k0, qf[ks], s0);
                s1 = AttMfma<TX>::run(k1, qf[ks], s1);
            }
        };
        // Plain attention (BIAS = false), every key of the tile in range, not the first tile: the accumulators START
        // at -m_run (the register block `negm`, rewritten only when the reference point moves), so the matrix pipe
        // delivers s - m_run and a weight is ONE v_exp_f32 -- no scale, no subtraction, no maximum.  The weights are
        // taken against the current reference point whatever this tile's maximum is: O and l carry the same factor,
        // the result is exact as long as nothing overflows -- which the row sum that is needed anyway tells: a lane
        // sum above ATT_LIMIT (or inf / NaN) sends the whole wave through the general path below, which recomputes
        // the tile's scores from LDS and moves the reference (first tiles and adversarial inputs only).
        bool general = BIAS || t == 0 || (t + 1) * ATT_BN > a.Nk;
        float lsum = 0.0f;
        if (!active) {
            general = false;  // a wave past the last query only helps staging the tiles
        } else if (!general) {
            scores(negm);
            v_fragments();
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                s0[v] = __builtin_amdgcn_exp2f(s0[v]);
                s1[v] = __builtin_amdgcn_exp2f(s1[v]);
            }
            {   // row sum as single f32 adds in four chains (packed v_pk_add_f32 costs more beside MFMAs than two adds)
                float c0 = att_add(s0[0], s1[0]), c1 = att_add(s0[1], s1[1]), c2 = att_add(s0[2], s1[2]),
                      c3 = att_add(s0[3], s1[3]);
#pragma unroll
                for (int v = 4; v < 16; v += 4) {
                    c0 = att_add(c0, s0[v]);     c1 = att_add(c1, s0[v + 1]);
                    c2 = att_add(c2, s0[v + 2]); c3 = att_add(c3, s0[v + 3]);
                    c0 = att_add(c0, s1[v]);     c1 = att_add(c1, s1[v + 1]);
                    c2 = att_add(c2, s1[v + 2]); c3 = att_add(c3, s1[v + 3]);
                }
                lsum = att_add(att_add(c0, c1), att_add(c2, c3));
            }
            general = !__all(lsum <= AttLimit<TX>::value);
            if (!general) l_run += lsum;
        }
        if (general) {
            att_f32x16 zero;
#pragma unroll
            for (int v = 0; v < 16; ++v) zero[v] = 0.0f;
            scores(zero);
            v_fragments();
            // ---- logits with the per-key bias (or the range mask); register v <-> key (v&3) + 8*(v>>2) + 4*hf (+32)
            float mt = -INFINITY;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b0 = *reinterpret_cast<const float4 *>(bias_row + 8 * g + 4 * hf);
                const float4 b1 = *reinterpret_cast<const float4 *>(bias_row + 32 + 8 * g + 4 * hf);
                s0[4 * g + 0] += b0.x; s0[4 * g + 1] += b0.y; s0[4 * g + 2] += b0.z; s0[4 * g + 3] += b0.w;
                s1[4 * g + 0] += b1.x; s1[4 * g + 1] += b1.y; s1[4 * g + 2] += b1.z; s1[4 * g + 3] += b1.w;
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) mt = fmaxf(mt, fmaxf(s0[v], s1[v]));
            {   // the partner lane l^32 holds the other half of this query's keys: v_permlane32_swap exchanges the two
                // halves in the vector ALU (no trip through the LDS pipeline in the middle of the softmax)
                const unsigned mb = __float_as_uint(mt);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const float m_new = fmaxf(m_run, mt);  // finite: every tile holds at least one key in range
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            lsum = 0.0f;
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
            }
            if (!BIAS) {
#pragma unroll
                for (int v = 0; v < 16; ++v) negm[v] = -m_new;
                asm volatile("" : "+v"(negm));  // a register block of its own, not a splat re-made per tile
            }
        }
        // ---- O^T += V^T P^T
        if (active) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                att_s16x8 pf, vf0, vf1;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = att_bits<TX>(kb == 0 ? s0[8 * p + e] : s1[8 * p + e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vf0[e] = vfr[kb][p][0][e];
                    vf0[4 + e] = vfr[kb][p][1][e];
                    vf1[e] = vfr[kb][p][2][e];
                    vf1[4 + e] = vfr[kb][p][3][e];
                }
                o0 = AttMfma<TX>::run(vf0, pf, o0);
                o1 = AttMfma<TX>::run(vf1, pf, o1);
            }
        }
        }
        __syncthreads();  // every wave is done with tile t
        if (t + 1 < ntiles) stage_write();
        __syncthreads();
    }

    // ---- out[b, q, h*64 + d] = O^T[d][q] / l ; register v <-> channel (v&3) + 8*(v>>2) + 4*hf (+32)
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (qrow < a.N) {
        short *op = reinterpret_cast<short *>(a.out) + b * a.o_sb + (int64_t)qrow * a.o_sn + h * a.o_sh + seg * a.o_seg;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            att_s16x4 w0, w1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w0[e] = att_bits<TX>(o0[4 * g + e] * inv);
                w1[e] = att_bits<TX>(o1[4 * g + e] * inv);
            }
            *reinterpret_cast<att_s16x4 *>(op + 8 * g + 4 * hf) = w0;
            *reinterpret_cast<att_s16x4 *>(op + 32 + 8 * g + 4 * hf) = w1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_attention_plain: softmax(q k^T * scale) v without a per-key term (VideoMAE's default, prop_attn = False --
// tome/patch/videomae.py:55-66 with size None; every patched attention before the first merge; the class token of
// Motionformer), software-pipelined over the key tiles:
//
//   iteration t:   S(t+1) = K(t+1) Q~^T - m_run     8 MFMAs   (score accumulators start at the block `negm`)
//                  O     += V(t)^T P(t)^T            8 MFMAs
//                  P(t+1) = exp2(S(t+1)), row sum, conversion to the 16-bit format    ~80 vector instructions
//
// The matrix work of an iteration does not depend on its vector work (P(t) was finished one iteration earlier), so
// the two interleave inside every wave -- the matrix pipe runs while the softmax issues -- instead of alternating
// between a matrix phase and a softmax phase that all waves of a workgroup enter together.  K/V tiles go through a
// three-slot LDS ring (tile t+1 is written while slow waves may still read V(t) and K(t+1)): one barrier per tile.
// First tile, a partly filled last tile, and a tile whose row sum trips the overflow guard take the general path
// (scores recomputed from zero, maximum, rescale of O and l) behind a wave-uniform branch.
// ------------------------------------------------------------------------------------------------
#define ATT_SLOTS 2
#ifndef ATT_ABL
#define ATT_ABL 0  // measurement builds only (tools/ab_lib.sh): bit 0 no v_exp, 1 no barrier, 2 no V-fragment reads,
#endif             // 3 no LDS staging writes, 4 no global loads in the fast step -- results are wrong by design

template <int V> struct AttInt { static constexpr int value = V; };

template <typename TX, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void k_attention_plain(AttnArgs a) {  // two waves per SIMD either way
    constexpr int ATT_BM = 32 * WAVES;
    // two LDS slots: tile t lives in slot t & 1.  Tile t+1 is written during iteration t, when every wave has left
    // iteration t-2 -- the last one that read slot (t+1) & 1 (K(t-1) for its scores, V(t-1) into registers).
    __shared__ __attribute__((aligned(16))) short lds_k[ATT_SLOTS][ATT_BN * ATT_KS];
    __shared__ __attribute__((aligned(16))) short lds_v[ATT_SLOTS][ATT_BN * ATT_VS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, hf = lane >> 5;
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

    const int qrow = qb * ATT_BM + wave * 32 + col;
    const int qload = qrow < a.N ? qrow : a.N - 1;
    const bool active = qb * ATT_BM + __builtin_amdgcn_readfirstlane(wave) * 32 < a.N;  // wave-uniform (scalar)
    const float sl = a.scale * 1.4426950408889634f;
    att_s16x8 qf[4];  // q * scale * log2(e), rounded once to the 16-bit format (see k_prop_attention)
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
    };
    // V^T fragments of a whole tile (layout: k_prop_attention); lane-constant base
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
    auto scores = [&](int S, const att_f32x16 &cinit) __attribute__((always_inline)) {
        // block 0 (keys 0..31) completes before block 1 starts: its weights can be taken while block 1 multiplies
        s0 = cinit;
        s1 = cinit;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            s0 = AttMfma<TX>::run(*reinterpret_cast<const att_s16x8 *>(kbase + S * ATT_BN * ATT_KS + 16 * ks), qf[ks], s0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            s1 = AttMfma<TX>::run(*reinterpret_cast<const att_s16x8 *>(kbase + S * ATT_BN * ATT_KS + 32 * ATT_KS + 16 * ks),
                                  qf[ks], s1);
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
    auto pv = [&]() __attribute__((always_inline)) {  // O^T += V(t)^T P(t)^T from the fragments in registers
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
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
    };
    // One iteration = tile t (weights and V fragments in registers) is accumulated while tile t+1 (slot S after the
    // barrier) gets its scores and weights.
    //   fast_step: tile t+1 holds 64 keys.  Straight-line code -- 16 MFMAs and ~80 vector instructions free to
    //   interleave.  The overflow guard only sets a flag (see below).
    bool bad = false;
    auto fast_step = [&](int S, int t, bool masked) __attribute__((always_inline)) {
#if !(ATT_ABL & 8)
        stage_write(S);  // registers hold tile t+1
#endif
        // tile t+2, rows clamped to the last key instead of a bounds branch (a partly filled tile gets its scores
        // masked by the general softmax; its weights are 0, so a repeated V row adds nothing): no control flow
        // between two barriers
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int key = min((t + 2) * ATT_BN + r0 + RSTEP * i, a.Nk - 1);
#if !(ATT_ABL & 16)
            kreg[i] = *reinterpret_cast<const uint4 *>(kp + (int64_t)key * a.k_sn + 8 * c0);
            vreg[i] = *reinterpret_cast<const uint4 *>(vp + (int64_t)key * a.v_sn + 8 * c0);
#endif
        }
#if !(ATT_ABL & 2)
        __syncthreads();   // tile t+1 visible; every wave has left iteration t-1
#endif
        // (measured: a hand-placed issue order of this block -- every MFMA followed by the vector work that fits its
        // shadow, pinned by sched_barrier fences -- runs within 1 % of what the compiler makes of it)
        scores(S, negm);
        pv();
#if !(ATT_ABL & 4)
        v_fragments(S);
#endif
#if !(ATT_ABL & 1)
#pragma unroll
        for (int v = 0; v < 16; ++v) s0[v] = __builtin_amdgcn_exp2f(s0[v]);
#pragma unroll
        for (int v = 0; v < 16; ++v) s1[v] = __builtin_amdgcn_exp2f(s1[v]);
#endif
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
        // iterations t = 0 .. ntiles-2 bring in tile t+1; those with t+1 < nfull may take the fast form
        int t = 0;
        if (active) {
            general_softmax(0, 0);
            v_fragments(0);
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
            pv();  // the last tile
        } else {
            for (; t + 1 < ntiles; ++t) helper_step(t);
        }
    }

    // ---- out[b, q, h*64 + d] = O^T[d][q] / l ; register v <-> channel (v&3) + 8*(v>>2) + 4*hf (+32)
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (qrow < a.N) {
        short *op = reinterpret_cast<short *>(a.out) + b * a.o_sb + (int64_t)qrow * a.o_sn + h * a.o_sh + seg * a.o_seg;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            att_s16x4 w0, w1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w0[e] = att_bits<TX>(o0[4 * g + e] * inv);
                w1[e] = att_bits<TX>(o1[4 * g + e] * inv);
            }
            *reinterpret_cast<att_s16x4 *>(op + 8 * g + 4 * hf) = w0;
            *reinterpret_cast<att_s16x4 *>(op + 32 + 8 * g + 4 * hf) = w1;
        }
    }
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

template <typename TX>
__global__ __launch_bounds__(256) void k_trajectory_mix(const TX *__restrict__ q2, const TX *__restrict__ k2,
                                                        const TX *__restrict__ val, int64_t rows, int S, int F, int H,
                                                        int64_t k_row, int64_t v_row, float scale,
                                                        TX *__restrict__ out, float *__restrict__ tattn) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= rows) return;  // wave-uniform
    const int C = H * 64, chunks = C >> 3;
    const TX *qr = q2 + row * C;
    const TX *kr = k2 + row * F * k_row;
    const TX *vr = val + row * F * v_row;
    float acc[2][8];
    float w[2][TRAJ_MAXF];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = lane + 64 * i;
        const bool on = c < chunks;
        float qv[8];
        if (on) load_pack<TX, 8>(qr + 8 * c, qv);
        uint4 kraw[TRAJ_MAXF];
#pragma unroll
        for (int f = 0; f < TRAJ_MAXF; ++f)
            if (on && f < F) kraw[f] = *reinterpret_cast<const uint4 *>(kr + (int64_t)f * k_row + 8 * c);
        float lg[TRAJ_MAXF];
#pragma unroll
        for (int f = 0; f < TRAJ_MAXF; ++f) {
            float d = 0.0f;
            if (on && f < F) {
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
            const int64_t b = row / S, s = row - b * S;
            float *tp = tattn + ((b * H + h) * S + s) * F;
            for (int f = 0; f < F; ++f) tp[f] = w[i][f];
        }
        uint4 vraw[TRAJ_MAXF];
#pragma unroll
        for (int f = 0; f < TRAJ_MAXF; ++f)
            if (on && f < F) vraw[f] = *reinterpret_cast<const uint4 *>(vr + (int64_t)f * v_row + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = 0.0f;
#pragma unroll
        for (int f = 0; f < TRAJ_MAXF; ++f) {
            if (on && f < F) {
                Pack<TX, 8> pk;
                __builtin_memcpy(&pk, &vraw[f], 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][e] = __builtin_fmaf(w[i][f], to_f32(pk.e[e]), acc[i][e]);
            }
        }
        if (on) store_pack<TX, 8>(out + row * C + 8 * c, acc[i]);
    }
}
