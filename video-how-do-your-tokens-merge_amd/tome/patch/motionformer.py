"""ToMe patch for Motionformer (reference: tome/patch/motionformer.py).  Trajectory attention over
1 + P*F tokens; the merge runs on F groups of P tokens obtained by the reference's own regrouping
'b (s f) d -> (b f) s d' (a stride-F slicing of the frame-major token sequence -- kept as it is, SURVEY.md
section 7.5), the class token is kept aside.  apply_patch takes the model itself."""
from __future__ import annotations

import os

import torch
from einops import rearrange

from . import _common as C
from .. import _abi
from ..merge import HeadMeanKeys

_KEYS_ONLY = os.environ.get("TOME_TRAJ_KEYS_ONLY", "1") != "0"  # 0 = evaluate all of proj_kv even when v2 is unused
_JOIN = os.environ.get("TOME_TRAJ_JOIN", "1") != "0"  # 0 = class row and trajectory rows through torch.cat (measurement switch)


def _block_forward(self, x, seq_len=196, num_frames=8, approx="none", num_landmarks=128):
    """ToMeBlock.forward (motionformer.py:15-30)."""
    info = self._tome_info
    attn_size = info["size"] if info["prop_attn"] else None
    attn_out, _, metric = self.attn(C.first_norm(self, x, info, self.norm1), seq_len=seq_len, num_frames=num_frames,
                                    approx=approx, num_landmarks=num_landmarks, size=attn_size, _want_attn=False)
    # x = x + attn; merge per group; norm2 -- one kernel when the layer merges 16-bit tokens
    x, y = C.merge_then_norm_regrouped(
        metric, x, info, self.norm2, lambda z: self.reduction_function(metric, z, info, num_frames),
        self.reduction_function is motionformer_merge, num_frames, residual=self.drop_path(attn_out),
        fold=C.foldable(self.mlp.fc2, not self.training) if C._plain_mlp(self.mlp) else None)
    return C.mlp_residual(self, self.mlp, x, y, info, drop_path=self.drop_path)


def qkv_attn(q, k, v):
    attn = torch.einsum("b i d, b j d -> b i j", q, k).softmax(dim=-1)
    return torch.einsum("b i j, b j d -> b i d", attn, v)


def _trajectory_forward(self, x, seq_len=196, num_frames=8, approx="none", num_landmarks=128,
                        size: torch.Tensor = None, _want_attn: bool = True):
    """ToMeTrajectoryAttention.forward (motionformer.py:33-144), exact attention only.  P is recomputed
    from the current token count; the size bias is added in the reference's '(s f)' order; the class
    token's own attention ignores size; metric = head-mean of the keys regrouped '(s f) -> (b f) s'.
    _want_attn=False (the patched block, which drops the temporal attention map): the map is not written."""
    if approx != "none":
        raise NotImplementedError("ToMe Motionformer patch: only approx='none' (the reference's ToMe configs)")
    B, N, _ = x.shape
    F = num_frames
    P = (N - 1) // F
    h = self.num_heads
    qkv = self.qkv(x)
    hd = qkv.shape[-1] // (3 * h)
    heads = qkv.view(B, N, 3, h, hd).permute(2, 0, 3, 1, 4)  # q, k, v as [B, h, N, hd] views of the projection
    # metric = rearrange(k_, '(b h) (s f) d -> (b f) h s d').mean(1) (motionformer.py:143-144): the regrouped keys
    # are a strided view of the qkv buffer -- [b, f, h, s, d] with token 1 + s*F + f -- and the head mean is taken
    # inside the matching kernel (tome_match_keys with inner groups), never as a tensor
    metric = HeadMeanKeys(heads[1][:, :, 1:1 + P * F, :].unflatten(2, (P, F)).permute(0, 3, 1, 2, 4))
    info = getattr(self, "_tome_info", None)
    ready = C.keys_ready(metric.keys, info, capture_only=True)  # the keys exist behind the qkv GEMM
    fused = (C._ATTN_KERNEL and not (self.training and self.attn_drop.p > 0.0) and all(_abi.prop_attention_ok(t) for t in heads))
    # flat per-key bias in the reference's '(s f)' order (motionformer.py:107-111): key j of the (f n)-ordered
    # sequence gets log(size) of (s = j // F, f = j % F)
    log_flat = None
    if size is not None:
        log_flat = rearrange(_abi.log_of_size(size), "(b f) s i -> b (s f) i", f=F)[:, :, 0]
    if fused:
        # the class token attends to every token, sizes ignored (motionformer.py:54): the attention kernel with one
        # query per head, read in place
        # (its row is row 0 of the buffer the temporal stage fills below: cat((cls_out, x), dim=1) without the copy)
        joined = torch.empty((B, N, h * hd), dtype=qkv.dtype, device=qkv.device) if _JOIN else None
        cls_out = _abi.prop_attention(heads[0][:, :, :1], heads[1], heads[2], None, self.scale,
                                      out=joined[:, :1].unflatten(2, (h, hd)) if _JOIN else None)
        # every token attends to the P keys of ONE frame at a time (softmax per frame): ONE launch of the segmented
        # attention kernel, queries and keys read in place from the qkv buffer, segment f writing its slice of
        # y 'b s f (h d)'; the [B*h, N, N] logits, their softmax and the attn @ v product never exist
        lf = None if log_flat is None else log_flat.float().contiguous()
        y = _abi.prop_attention_segments(heads[0][:, :, 1:], heads[1][:, :, 1:1 + P * F], heads[2][:, :, 1:1 + P * F], F,
                                         self.scale, log_bias=lf)
    else:
        q, k, v = (rearrange(t, "b n (h d) -> (b h) n d", h=h) for t in qkv.chunk(3, dim=-1))
        (cls_q, q_), (cls_k, k_), (cls_v, v_) = ((t[:, 0:1], t[:, 1:]) for t in (q, k, v))
        cls_out = rearrange(qkv_attn(cls_q * self.scale, k, v), "(b h) f d -> b f (h d)", f=1, h=h)
        q_dot_k = rearrange(q_ @ k_.transpose(-2, -1), "b q (f n) -> b q f n", f=F) * self.scale
        if log_flat is not None:
            q_dot_k = rearrange(q_dot_k, "(b h) q f n -> b h q (f n)", h=h, f=F)
            q_dot_k = q_dot_k + log_flat[:, None, None, :].to(q_dot_k.dtype)
            q_dot_k = rearrange(q_dot_k, "b h q (f n) -> (b h) q f n", h=h, f=F)
        attn = self.attn_drop(q_dot_k.softmax(dim=-1))
        v_ = rearrange(v_, "b (f n) d -> b f n d", f=F, n=P)
        y = torch.einsum("b q f n, b f n d -> b q f d", attn, v_)
        y = rearrange(y, "(b h) s f d -> b s f (h d)", b=B)
    C.match_beside(metric, ready, info)  # inside a graph capture: the matching on the side stream, beside the temporal stage
    y_diag = rearrange(y, "b (g n) f d -> b g n f d", g=F)
    y_diag = torch.diagonal(y_diag, dim1=-4, dim2=-2)
    y_diag = rearrange(y_diag, "b n d f -> b (f n) d", f=F)
    q2p = self.proj_q(y_diag)  # [B, S, C]
    Cc = q2p.shape[-1]
    if self.use_original_code and _KEYS_ONLY and C._stock_module(self.proj_kv, torch.nn.Linear):
        # use_original_code (the reference's default): the weighted sum runs over the trajectory tokens y themselves,
        # v2 -- the second half of proj_kv's output -- is computed and never read (motionformer.py:123,130-134).  Only
        # the key half of the projection is evaluated: half of the model's largest GEMM ([B*S*F, C] x [C, 2C]).
        pb = self.proj_kv.bias
        k2_tok = torch.nn.functional.linear(y, self.proj_kv.weight[:Cc], None if pb is None else pb[:Cc])
        val_tok = y
    else:
        kv = self.proj_kv(y)       # [B, S, F, 2C]: keys | values
        k2_tok = kv[..., :Cc]
        val_tok = y if self.use_original_code else kv[..., Cc:]
    if fused and _abi.trajectory_mix_ok(q2p, k2_tok, val_tok, h):
        # F logits per (token, head), their softmax and the weighted sum of the F trajectory tokens: one streaming
        # pass over k2 and val (tome_trajectory_mix) instead of two multiplies, two reductions and a softmax
        out, tattn = _abi.trajectory_mix(q2p, k2_tok, val_tok, h, self.scale, want_attn=_want_attn,
                                         out=joined[:, 1:] if _JOIN else None)
        if tattn is not None:
            tattn = tattn.to(x.dtype)
        # class row + trajectory rows, already side by side
        out = joined if _JOIN else torch.cat((cls_out.reshape(B, 1, -1), out), dim=1)
    else:
        q2 = rearrange(q2p, "b s (h d) -> b h s d", h=h) * self.scale
        k2 = rearrange(k2_tok, "b s f (h d) -> b h s f d", f=F, h=h)
        # F = 8 logits per trajectory: a broadcast multiply + reduction streams k2 once; as the batched
        # [1 x d] @ [d x F] products the einsum form lowers to, it is the slowest kernel of the model on MI355X
        tattn = (k2 * q2.unsqueeze(-2)).sum(dim=-1).softmax(dim=-1)
        val = rearrange(val_tok, "b s f (h d) -> b h s f d", f=F, h=h)
        out = rearrange((val * tattn.unsqueeze(-1)).sum(dim=-2), "b h s d -> b s (h d)")  # same remark
        out = torch.cat((cls_out.reshape(B, 1, -1), out), dim=1)
    out = self.proj_drop(self.proj(out))
    return out, tattn, metric


def _regroup(x, num_frames):
    return rearrange(x[:, 1:, :], "b (s f) d -> (b f) s d", f=num_frames)


def _ungroup(cls, y, num_frames):
    return torch.cat((cls, rearrange(y, "(b f) s d -> b (s f) d", f=num_frames)), dim=1)


def motionformer_merge(metric, x, _tome_info, num_frames):
    r = _tome_info["r"].pop(0)
    if r > 0:
        # 'b (s f) d -> (b f) s d', merge, back, cls in front: by the kernel's addressing, no copies
        x = C.reduce_merge_regrouped(metric, x, _tome_info, r, num_frames)
    return x


def motionformer_drop(metric, x, _tome_info, num_frames):
    r = _tome_info["r"].pop(0)
    if r > 0:
        x = C.reduce_drop_regrouped(metric, x, _tome_info, r, num_frames)  # groups addressed in place, no permuted copies
    return x


def motionformer_hybrid(metric, x, _tome_info, num_frames):
    r = _tome_info["r"].pop(0)
    if r > 0:
        x = C.reduce_merge_regrouped(metric, x, _tome_info, r, num_frames, hybrid=True)
    return x


def apply_duplicate_patch(model, layer_to_duplicate, quantity):
    """motionformer.py:230-232: the same block object is visited `quantity` times."""
    for i in range(layer_to_duplicate + 1, layer_to_duplicate + quantity):
        model.blocks.insert(index=i, module=model.blocks[layer_to_duplicate])


def _is_block(m) -> bool:
    return all(hasattr(m, a) for a in ("attn", "mlp", "norm1", "norm2")) and hasattr(m.attn, "proj_kv")


def _is_trajectory_attention(m) -> bool:
    return all(hasattr(m, a) for a in ("qkv", "proj_q", "proj_kv", "proj", "use_original_code"))


def apply_patch(model, trace_source: bool = False, prop_attn: bool = True, mode: str = "merge",
                head_aggregation: str = "mean", threshold: float = 0.0, verbose: bool = False):
    C.wrap_model_forward(model, lambda w: w.blocks)
    model.r = 0
    info = C.new_tome_info(trace_source, prop_attn, mode, head_aggregation, threshold, verbose, class_token=False)
    del info["head_aggregation"]  # not part of the reference's Motionformer dict (motionformer.py:254-265)
    model._tome_info = info
    reduction_function = C.pick_reduction(mode, motionformer_merge, motionformer_drop, motionformer_hybrid)
    for module in model.modules():
        if _is_block(module):
            C.swizzle(module, "ToMeBlock", {"forward": _block_forward})
            module._tome_info = model._tome_info
            module.reduction_function = reduction_function
        elif _is_trajectory_attention(module):
            C.swizzle(module, "ToMeTrajectoryAttention", {"forward": _trajectory_forward})
            module._tome_info = model._tome_info  # (the layer's r: its matching may start beside its attention)
    C.link_next_norms(model.blocks, "norm1")
