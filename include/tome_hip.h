/*
 * tome_hip.h -- C ABI of the MI355X (gfx950) ToMe merge path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  Every entry point is
 * asynchronous on the caller's HIP stream, allocates nothing, never synchronises and returns a
 * status (0 = ok).  All pointers are DEVICE pointers owned by the caller.
 *
 * The reference (sjpollard/video-how-do-your-tokens-merge) is pure Python/PyTorch, so its "FFI"
 * for this path is the function interface of tome/merge.py; each entry below names the reference
 * lines it replaces.  INTEGRATION.md shows the ctypes binding a maintainer of the reference would
 * add (it is the binding video-how-do-your-tokens-merge_amd/tome/_abi.py uses).
 *
 * Index conventions (merge.py:52,64-73): tokens with even position are set A ("src
 * candidates", row i = t/2, T1 = ceil(T/2) rows), odd positions are set B ("dst", row j = t/2,
 * T2 = floor(T/2) rows).  src_idx / unm_idx hold A rows, dst_idx holds B rows, all int64 like the
 * reference's closure variables ([n,r,1] / [n,T1-r,1], trailing 1 implicit here).
 */
#ifndef TOME_HIP_H
#define TOME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *tome_stream_t; /* hipStream_t */

enum tome_dtype { TOME_F32 = 0, TOME_BF16 = 1, TOME_F16 = 2 };

/* reduce argument of torch.Tensor.scatter_reduce as used by merge(x, mode) -- merge.py:80 */
enum tome_mode { TOME_SUM = 0, TOME_MEAN = 1, TOME_AMAX = 2, TOME_PROD = 3, TOME_AMIN = 4 };

enum tome_status {
    TOME_OK = 0,
    TOME_EINVAL = 1,     /* bad argument (shape, dtype, null pointer, misaligned index buffer) */
    TOME_EWORKSPACE = 2, /* workspace smaller than tome_match_workspace_bytes() */
    TOME_ELAUNCH = 3     /* HIP reported a launch error (text in tome_last_error()) */
};

#define TOME_ABI_VERSION 9

int tome_abi_version(void);

/* Thread-local text of the last non-zero status returned on this thread. */
const char *tome_last_error(void);

/* merge.py:36-47 -- r clamped to half of the unprotected tokens; <= 0 means "do nothing". */
int64_t tome_effective_r(int64_t T, int64_t r, int class_token, int distill_token);

/* Bytes of scratch tome_match needs for a [n,T,D] metric. */
size_t tome_match_workspace_bytes(int64_t n, int64_t T, int64_t D);

/*
 * tome_match  <-  bipartite_soft_matching, index part (merge.py:49-73; identical code in
 *                 bipartite_soft_matching_drop :236-251 and _hybrid :296-311).
 *
 * metric: [n,T,D] of `dtype`, element strides (stride_n, stride_t, 1) -- a strided view such as
 *         timesformer.py:83 `k.mean(1)[:, 1:, :]` needs no copy.
 * r:      the caller's r; the call clamps it itself (tome_effective_r) and the index buffers must
 *         be sized for the clamped value: src_idx, dst_idx [n, r_eff], unm_idx [n, T1 - r_eff].
 * node_max: optional [n,T1] fp32 (row maxima of the similarity matrix, merge.py:64; needed by
 *         the hybrid threshold test merge.py:326).
 * row_map:  optional [n,T1] int32: for every A row the row of the MERGED sequence it ends up in
 *         (its own slot when unmerged, its destination's slot when merged).
 * Arithmetic: metric is converted to fp32; unit vectors, the A.B^T similarity (fp32 MFMA, k-ordered
 *         fma chain), first-index row argmax and a stable descending ranking -- the contract is
 *         written out in oracle/tome_oracle.c and DESIGN.md.
 * Returns TOME_OK also when r_eff <= 0 (nothing is written).
 */
int tome_match(const void *metric, int dtype, int64_t n, int64_t T, int64_t D, int64_t stride_n,
               int64_t stride_t, int64_t r, int class_token, int distill_token, int64_t *src_idx,
               int64_t *dst_idx, int64_t *unm_idx, float *node_max, int32_t *row_map,
               void *workspace, size_t workspace_bytes, tome_stream_t stream);

/*
 * tome_match_keys  <-  the metric producer fused in front of tome_match:
 *     metric = k.mean(1)   (tome/patch/videomae.py:72-73, timesformer.py:83, motionformer.py:143-144,
 *     vivit.py:123-124), then bipartite_soft_matching(metric, ...) (merge.py:49-73).
 *
 * keys: per-head attention keys [n,H,T,D] of `dtype`, element strides (stride_n, stride_h, stride_t, 1) --
 *       e.g. the k slice of the qkv projection buffer, no copy; every stride and the base must be 16-byte
 *       aligned; D must be 64.  The head mean is taken as torch does on CPU: fp32 sum in head order, one
 *       division by H, one rounding to `dtype`; everything after is tome_match.
 * inner, stride_inner: groups interleaved inside a clip (Motionformer's '(b h) (s f) d -> (b f) h s d',
 *       motionformer.py:143-144): group g = o * inner + f starts at o * stride_n + f * stride_inner, its tokens are
 *       stride_t apart (inner token rows).  inner = 1 (stride_inner ignored): n independent [H,T,D] blocks.
 */
int tome_match_keys(const void *keys, int dtype, int64_t n, int64_t H, int64_t T, int64_t D, int64_t stride_n,
                    int64_t inner, int64_t stride_inner, int64_t stride_h, int64_t stride_t, int64_t r,
                    int class_token, int distill_token,
                    int64_t *src_idx, int64_t *dst_idx, int64_t *unm_idx, float *node_max, int32_t *row_map,
                    void *workspace, size_t workspace_bytes, tome_stream_t stream);

/*
 * tome_match_scores  <-  the same selection from caller-made scores [n,T1,T2] fp32
 *                        (merge.py:54-57 random_merge / :239-242 random_drop, scores = torch.rand).
 */
int tome_match_scores(const float *scores, int64_t n, int64_t T, int64_t r, int class_token,
                      int distill_token, int64_t *src_idx, int64_t *dst_idx, int64_t *unm_idx,
                      float *node_max, int32_t *row_map, void *workspace, size_t workspace_bytes,
                      tome_stream_t stream);

/* merge.py:326 -- edge_keep[n,r] = (node_max[src_idx[k]] >= threshold) as 0/1 bytes. */
int tome_edge_keep(const float *node_max, const int64_t *src_idx, int64_t n, int64_t T, int64_t r,
                   float threshold, uint8_t *edge_keep, tome_stream_t stream);

/*
 * tome_merge_wavg  <-  merge_wavg (merge.py:355-369) fused: x*size, the two "sum" merges, x/size.
 *
 * x [n,T,C] of x_dtype, contiguous.  size: NULL (ones, :362-363) or [n,T] of size_dtype.
 * x_out [n,T-r,C] of x_dtype, size_out [n,T-r] of size_dtype.  r is the clamped r (> 0).
 * edge_keep: NULL, or the hybrid flags (merge.py:326).
 * log_size_out: NULL, or [n,T-r] of size_dtype receiving log(size_out) -- the proportional-attention bias the
 * next block adds to its logits (`size.log()`, tome/patch/videomae.py:62-63, timesformer.py:73-74,
 * motionformer.py:107-111, vivit.py:103-104): fp32 log of the stored size, rounded to size_dtype.  All four
 * tome_merge_wavg* entries take it.
 * Arithmetic in fp32 (products, then sequential adds from the destination's own term in src_idx
 * order, then one division); results rounded once to the output dtype.
 */
int tome_merge_wavg(const void *x, int x_dtype, const void *size, int size_dtype, int64_t n,
                    int64_t T, int64_t C, int64_t r, const int64_t *src_idx,
                    const int64_t *dst_idx, const int64_t *unm_idx, int distill_token,
                    const uint8_t *edge_keep, void *x_out, void *size_out, void *log_size_out,
                    tome_stream_t stream);

/*
 * tome_merge_wavg_ln  <-  merge_wavg followed by the block's second LayerNorm:
 *     x = merge_wavg(merge, x, size); ... self.norm2(x)      (tome/patch/videomae.py:25-27, vivit.py:38-41)
 * x_out as tome_merge_wavg; y_out [n,T-r,C] = LayerNorm(x_out) over the channels (weight, bias [C] of the token
 * dtype, eps), computed in fp32 from the stored x_out and rounded once -- so the MLP reads y_out and the
 * separate LayerNorm pass over the merged tokens disappears.  16-bit tokens, C <= 1024, C % 8 == 0.
 * addend: NULL, or [n,T,C] like x: the tokens that are merged are round_to_dtype(x + addend), i.e. the residual
 * `x = x + attn(norm1(x))` in front of the merge (videomae.py:20, vivit.py:35) is taken while loading and the
 * separate add pass disappears as well.
 * x_out_bias: NULL, or [C] of the token dtype: x_out is stored as round(x' + x_out_bias) while y_out stays
 * LayerNorm(x').  For callers that let the MLP's second GEMM accumulate onto x_out in place
 * (`x = x + self.mlp(self.norm2(x))`, videomae.py:29, as `x_out.addmm_(h, W2^T)`): that GEMM's bias is in the buffer
 * beforehand, and the block's second residual add is no pass of its own.
 */
int tome_merge_wavg_ln(const void *x, int x_dtype, const void *size, int size_dtype, int64_t n, int64_t T,
                       int64_t C, int64_t r, const int64_t *src_idx, const int64_t *dst_idx,
                       const int64_t *unm_idx, int distill_token, const uint8_t *edge_keep,
                       const void *ln_weight, const void *ln_bias, float eps, const void *addend, void *x_out,
                       void *y_out, void *size_out, void *log_size_out, const void *x_out_bias,
                       tome_stream_t stream);

/*
 * tome_merge_wavg_regrouped  <-  the rearrange / merge_wavg / rearrange / cat sequence of
 *     timesformer_merge (tome/patch/timesformer.py:89-107) and motionformer_merge
 *     (tome/patch/motionformer.py:150-168), without the two permuted copies of x.
 *
 * x [B, has_cls + P*F, C]: token `has_cls + p*F + f` of clip b belongs to merge group b*F + f (n = B*F groups
 * of P tokens, the layout both patches regroup into); the index buffers and size [n,P] / size_out [n,P-r]
 * are those of the n groups.  x_out [B, has_cls + (P-r)*F, C] in the same interleaved layout, class token
 * rows copied through.  Arithmetic as tome_merge_wavg.  C * sizeof(dtype) must be a multiple of 16.
 */
int tome_merge_wavg_regrouped(const void *x, int x_dtype, const void *size, int size_dtype, int64_t B,
                              int64_t F, int64_t P, int64_t C, int64_t r, int has_cls,
                              const int64_t *src_idx, const int64_t *dst_idx, const int64_t *unm_idx,
                              const uint8_t *edge_keep, void *x_out, void *size_out, void *log_size_out,
                              tome_stream_t stream);

/* tome_merge_wavg_regrouped with the residual add in front and the block's norm2 behind it fused in, as
 * tome_merge_wavg_ln does for the plain layout (timesformer.py:52-56, motionformer.py:24-29).  The class-token
 * rows get the same add + LayerNorm.  16-bit tokens, C <= 1024.
 * addend: NULL, or the residual in x's layout [B, has_cls + P*F, C] (addend_grouped = 0), or -- addend_grouped = 1 --
 * in the GROUPED layout [B*F, has_cls + P, C] the spatial attention of TimeSformer leaves it in
 * (tome/patch/timesformer.py:32-52: `res_spatial`, whose '(b t) (h w) m -> b (h w t) m' rearrangement and `cat` with
 * the frame-averaged class token are then not made); its class rows are ignored and the class tokens' addend is
 * cls_addend [B, C] (NULL: none).  x_out_bias: as in tome_merge_wavg_ln (class-token rows included). */
int tome_merge_wavg_regrouped_ln(const void *x, int x_dtype, const void *size, int size_dtype, int64_t B,
                                 int64_t F, int64_t P, int64_t C, int64_t r, int has_cls,
                                 const int64_t *src_idx, const int64_t *dst_idx, const int64_t *unm_idx,
                                 const uint8_t *edge_keep, const void *ln_weight, const void *ln_bias, float eps,
                                 const void *addend, int addend_grouped, const void *cls_addend, void *x_out,
                                 void *y_out, void *size_out, void *log_size_out, const void *x_out_bias,
                                 tome_stream_t stream);

/* tome_drop (below) on the regrouped layout of tome_merge_wavg_regrouped: timesformer_drop / motionformer_drop
 * (tome/patch/timesformer.py:111-131, motionformer.py:172-193) without the permuted copies.
 * x [B, has_cls + P*F, C] -> x_out [B, has_cls + (P-r)*F, C]; und_idx [B*F, ceil(P/2)-r].
 * C * sizeof(dtype) must be a multiple of 16. */
int tome_drop_regrouped(const void *x, int dtype, int64_t B, int64_t F, int64_t P, int64_t C, int64_t r,
                        int has_cls, const int64_t *und_idx, void *x_out, tome_stream_t stream);

/*
 * tome_prop_attention  <-  the proportional attention of the patched attention modules:
 *     attn = softmax(q k^T * scale + size.log()[:, None, None, :, 0]) ; x = attn @ v
 *     (ToMeAttention.forward, tome/patch/videomae.py:55-66; vivit.py:95-113; timesformer.py:66-78 with the bias
 *      on the non-class block only: bias_skip = 1)
 * q: [B, H, N, 64], k, v: [B, H, Nk, 64] views of 16-bit tensors, element strides {batch, head, token} each (channels
 * contiguous, rows 16-byte aligned) -- e.g. the three slices of a [B, N, 3, H, 64] qkv buffer, read in place.  Nk may
 * differ from N (Motionformer's trajectory attention attends from every token to the keys of one frame at a time,
 * tome/patch/motionformer.py:98-121).
 * log_size: NULL (plain attention) or fp32 [B, Nk - bias_skip] with row stride log_size_stride: log of the token
 * sizes, added to the logits of key j (bias_skip = 1, N == Nk: key 0 and query 0 carry no bias, entry j-1 belongs
 * to key j).
 * out: NULL strides -> [B, N, H*64] contiguous, the layout the output projection reads; else element strides
 * {batch, head, token} of out[b, q, h, 0..63] (rows 8-byte aligned).  fp32 softmax and accumulation.
 */
int tome_prop_attention(const void *q, const void *k, const void *v, int dtype, int64_t B, int64_t H, int64_t N,
                        int64_t Nk, int64_t D, const int64_t *q_strides, const int64_t *k_strides,
                        const int64_t *v_strides, const float *log_size, int64_t log_size_stride, int bias_skip,
                        float scale, void *out, const int64_t *out_strides, tome_stream_t stream);

/* tome_prop_attention_segments  <-  the per-frame stage of ToMeTrajectoryAttention.forward
 * (tome/patch/motionformer.py:98-121): every query attends to the keys of ONE frame at a time -- `nseg` independent
 * key ranges of Nk keys each with their own softmax -- in ONE launch.  Segment s reads k + s*seg_strides[0],
 * v + s*seg_strides[1], log_size + s*seg_strides[3] and writes out + s*seg_strides[2] (element offsets); the
 * other arguments are tome_prop_attention's (no bias_skip form).  The [B*H, N, nseg*Nk] logits never exist. */
int tome_prop_attention_segments(const void *q, const void *k, const void *v, int dtype, int64_t B, int64_t H,
                                 int64_t N, int64_t Nk, int64_t D, const int64_t *q_strides,
                                 const int64_t *k_strides, const int64_t *v_strides, const float *log_size,
                                 int64_t log_size_stride, float scale, void *out, const int64_t *out_strides,
                                 int64_t nseg, const int64_t *seg_strides, tome_stream_t stream);

/*
 * tome_trajectory_mix  <-  the temporal stage of ToMeTrajectoryAttention.forward (tome/patch/motionformer.py:122-139):
 *     attn = softmax(einsum('b h s d, b h s f d -> b h s f', q2 * scale, k2)); x = einsum('b h s f, b h s f d -> b h s d', attn, v2)
 * q2 [B, S, H*64]; k2, val [B, S, F, H*64] views whose (b, s, f) rows are k_row_stride / v_row_stride elements apart
 * (k2 = first half of the proj_kv output, val = the trajectory tokens or its second half); out [B, S, H*64] rows,
 * batch b starting out_batch_stride elements after batch b-1 (0 = S*H*64, contiguous; larger: `out` is a slice of
 * the [B, 1+S, C] buffer that also takes the class token's row, so that `torch.cat((cls_out, x), dim=1)` of
 * motionformer.py:138 needs no copy); tattn NULL or fp32 [B, H, S, F].  16-bit tensors, head dim 64, H <= 16,
 * F <= 8; fp32 arithmetic inside.
 */
int tome_trajectory_mix(const void *q2, const void *k2, const void *val, int dtype, int64_t B, int64_t S, int64_t F,
                        int64_t H, int64_t D, int64_t k_row_stride, int64_t v_row_stride, float scale, void *out,
                        int64_t out_batch_stride, float *tattn, tome_stream_t stream);

/*
 * tome_short_attention  <-  `self.temporal_attn(...)` in ToMeBlock.forward of the TimeSformer patch
 * (tome/patch/timesformer.py:25-27): the host model's attention over the T <= 8 copies of one spatial token,
 *     softmax(q k^T * scale) v        per (sequence, head), sequences = 'b (p t) m -> (b p) t m'
 * q, k, v: [B, H, N, 64] views ({batch, head, token} element strides; the heads of a token side by side: head
 * stride 64 -- the slices of a qkv projection read in place); out [B, N, H*64] contiguous.  16-bit tensors,
 * N <= 8, fp32 arithmetic inside.  No bias term: the temporal attention never sees token sizes.
 */
int tome_short_attention(const void *q, const void *k, const void *v, int dtype, int64_t B, int64_t H, int64_t N,
                         int64_t D, const int64_t *q_strides, const int64_t *k_strides, const int64_t *v_strides,
                         float scale, void *out, tome_stream_t stream);

/*
 * tome_add_layernorm  <-  the second residual of the patched block and the LayerNorm that consumes it:
 *     x = x + self.drop_path(self.mlp(self.norm2(x)))      (tome/patch/videomae.py:29)
 *     ... next ToMeBlock.forward: self.norm1(x)            (tome/patch/videomae.py:19)
 * x_out = round(x + addend), y_out = LayerNorm(x_out), both [rows, C] of `dtype` (16-bit, C <= 1024, C % 8 == 0).
 * addend NULL: y_out = LayerNorm(x) only -- x already holds the sum (the MLP's GEMM accumulated onto it, see
 * x_out_bias of tome_merge_wavg_ln); x_out is then ignored and may be NULL.
 */
int tome_add_layernorm(const void *x, const void *addend, int dtype, int64_t rows, int64_t C,
                       const void *ln_weight, const void *ln_bias, float eps, void *x_out, void *y_out,
                       tome_stream_t stream);

/* tome_add_layernorm for rows that come in `groups` groups of `group_rows` whose first row is a class token the
 * consumer of y does not read: x_out [groups*group_rows, C] as before, y_out [groups*(group_rows-1), C] holds the
 * LayerNorm of the other rows, compacted.  TimeSformer: `self.temporal_norm1(xt)` with `xt = x[:, 1:, :]`
 * (tome/patch/timesformer.py:24-26) -- the regrouping '(b p) t m' of y is then a view. */
int tome_add_layernorm_skip_first(const void *x, const void *addend, int dtype, int64_t groups, int64_t group_rows,
                                  int64_t C, const void *ln_weight, const void *ln_bias, float eps, void *x_out,
                                  void *y_out, tome_stream_t stream);

/*
 * tome_add_layernorm_regrouped  <-  the middle of TimeSformer's divided space-time ToMeBlock.forward
 * (tome/patch/timesformer.py:24-38): the temporal attention's residual, the regrouping of the tokens for the
 * spatial attention (class token replicated into every frame) and that attention's LayerNorm:
 *     xt = x[:, 1:, :] + res_temporal;  x1 = cat(cls, xt)
 *     xs = cat(cls repeated per frame, rearrange(xt, 'b (p t) m -> (b t) p m'));  y = self.norm1(xs)
 * x [B, 1 + P*F, C], addend [B, P*F, C] -> x_out = x1 [B, 1 + P*F, C] and y_out = norm1(xs) [B*F, 1 + P, C]
 * (16-bit, C <= 1024, C % 8 == 0): one pass, the regrouped un-normalised copy `xs` is never written.
 */
int tome_add_layernorm_regrouped(const void *x, const void *addend, int dtype, int64_t B, int64_t F, int64_t P,
                                 int64_t C, const void *ln_weight, const void *ln_bias, float eps, void *x_out,
                                 void *y_out, tome_stream_t stream);

/* tome_merge  <-  merge(x, mode) closure (merge.py:75-85; hybrid :313-334 when edge_keep). */
int tome_merge(const void *x, int dtype, int64_t n, int64_t T, int64_t C, int64_t r,
               const int64_t *src_idx, const int64_t *dst_idx, const int64_t *unm_idx,
               int distill_token, int mode, const uint8_t *edge_keep, void *out,
               tome_stream_t stream);

/* tome_drop  <-  drop(x) closure (merge.py:253-262): unmerged A rows then all B rows. */
int tome_drop(const void *x, int dtype, int64_t n, int64_t T, int64_t C, int64_t r,
              const int64_t *und_idx, int distill_token, void *out, tome_stream_t stream);

/* tome_unmerge  <-  unmerge(x) closure (merge.py:87-100): x [n,T-r,C] -> out [n,T,C]. */
int tome_unmerge(const void *x, int dtype, int64_t n, int64_t T, int64_t C, int64_t r,
                 const int64_t *src_idx, const int64_t *dst_idx, const int64_t *unm_idx, void *out,
                 tome_stream_t stream);

/* tome_gelu_erf  <-  the activation of the MLP the patched block calls between merge and second residual
 * (`x = x + self.drop_path(self.mlp(self.norm2(x)))`, tome/patch/videomae.py:29, timesformer.py:56,
 * motionformer.py:29; the models' `act_layer=nn.GELU`): y = x * 0.5 * (1 + erf(x / sqrt(2))) on `elements` 16-bit
 * values (a multiple of 8), fp32 arithmetic, bit-identical to the framework's kernel.  y may alias x. */
int tome_gelu_erf(const void *x, int dtype, int64_t elements, void *y, tome_stream_t stream);

/* tome_tubelet_rows  <-  the models' patch embedding, a convolution whose stride equals its kernel
 * (slowfast/models/videomae_video_model_builder.py:137-166 `PatchEmbed.proj`; TimeSformer's per-frame Conv2d; Motionformer
 * `PatchEmbed3D`; ViViT's tubelet Conv3d): its input side as the [B*N, C*kt*kh*kw] matrix the weight multiplies,
 *     rows[b, (t', h', w'), (c, dt, dh, dw)] = x[b, c, t'*kt + dt, h'*kh + dh, w'*kw + dw]      (a pure move)
 * x: any view [B, C, T, H, W] with unit stride along W, x_strides = element strides {b, c, t, h}; elem_bytes 2 or 4;
 * kw * elem_bytes and every stride a multiple of 16 bytes.  Token order (t', h', w') row-major =
 * `conv(x).flatten(2).transpose(1, 2)`; inner order = the flattened convolution weight's. */
int tome_tubelet_rows(const void *x, int elem_bytes, int64_t B, int64_t C, int64_t T, int64_t H, int64_t W,
                      const int64_t *x_strides, int64_t kt, int64_t kh, int64_t kw, void *rows,
                      tome_stream_t stream);

/* tome_row_map / tome_source_init  <-  merge_source(merge, x, source=None) (merge.py:372-384) and the drop modes'
 * `drop(eye)` (tome/patch/videomae.py:112-117): the first layer's source matrix.  The reference builds an
 * [n,T,T] identity and merges it with mode "max"; element (o, t) of the result is 1 exactly when token t lands in
 * merged row o, i.e. the matching's row map -- so the identity is never made.
 *   tome_row_map      row_map [n,T1] int32 from the index tensors (same values tome_match writes when asked to)
 *   tome_source_init  source_out [n,T-r,T] fp32, one coalesced pass; drop != 0: merged-away tokens get no row
 */
int tome_row_map(int64_t n, int64_t T, int64_t r, int distill_token, const int64_t *src_idx,
                 const int64_t *dst_idx, const int64_t *unm_idx, int32_t *row_map, tome_stream_t stream);
int tome_source_init(int64_t n, int64_t T, int64_t r, int distill_token, int drop, const int32_t *row_map,
                     float *source_out, tome_stream_t stream);

#ifdef TOME_PROFILE_HOOKS
/*
 * MEASUREMENT BUILD ONLY (lib/libtome_hip_prof.so, compiled with -DTOME_PROFILE_HOOKS; not part of the product
 * ABI, not in libtome_hip.so): tome_profile_enable(reps > 0) makes tome_match on the calling thread record HIP
 * events between its stages on the caller's stream and launch every stage kernel `reps` times back to back (the
 * kernels are pure functions of their inputs, results are unchanged); tome_profile_read waits for the last
 * profiled call and returns the milliseconds PER LAUNCH of the stages {unit vectors, similarity+row max,
 * rank+select}.  bench.py's stage-timing leg is the only caller.  No reference counterpart.
 */
int tome_profile_enable(int on);
int tome_profile_read(float *stage_ms, int max_stages);
#endif

#ifdef __cplusplus
}
#endif
#endif /* TOME_HIP_H */
