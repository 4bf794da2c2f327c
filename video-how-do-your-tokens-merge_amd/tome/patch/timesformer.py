"""ToMe patch for TimeSformer (reference: tome/patch/timesformer.py).  Divided space-time attention: the
merge runs per frame on the spatial tokens ('b (p t) m -> (b t) p m'), the class token is kept aside, every
frame loses the same r tokens so the frame groups stay rectangular."""
from __future__ import annotations

import torch

from . import _common as C
from .. import _abi
from ..merge import HeadMeanKeys


def _block_forward(self, x, B, T, W):
    """ToMeBlock.forward (timesformer.py:13-57)."""
    info = self._tome_info
    attn_size = info["size"] if info["prop_attn"] else None
    if self.attention_type in ("space_only", "joint_space_time"):
        x = x + self.drop_path(self.attn(self.norm1(x))[0])
        return x + self.drop_path(self.mlp(self.norm2(x)))
    P = (x.size(1) - 1) // T  # spatial tokens per frame right now (H and W mean nothing after merging)
    m = x.size(2)
    # temporal attention over the T copies of every spatial token
    # temporal_norm1 is per row: taken over the whole token tensor (one extra class row per clip) it can come
    # from the previous block's fused residual + LayerNorm (finish_block below)
    # (only the patch tokens of it are read: the hand-over leaves the class row out, so the regrouping is a view)
    xn = C.first_norm(self, x, info, self.temporal_norm1, skip_first=True)
    rt = self.drop_path(self.temporal_attn(xn.reshape(B * P, T, m))).reshape(B, P * T, m)
    cls0 = x[:, :1, :]
    rt = self.temporal_fc(rt)
    if C._FUSE_NEXT and rt.dtype == x.dtype and _abi.ln_fusable(x, self.norm1):
        # residual of the temporal attention, 'b (p t) -> (b t) p' with the class token in front of every frame,
        # and the spatial attention's norm1 -- one pass (tome_add_layernorm_regrouped); the regrouped
        # un-normalised tokens never exist
        x1, xs_normed = _abi.add_layernorm_regrouped(x, rt, T, self.norm1.weight, self.norm1.bias, self.norm1.eps)
    else:
        if torch.is_grad_enabled() and x.requires_grad:
            # the reference's op sequence (differentiable): add, transpose, three cats
            xt = x[:, 1:, :] + rt
            x1 = torch.cat((cls0, xt), 1)
            xs_in = torch.cat((cls0.expand(B, T, m).reshape(B * T, 1, m),
                               xt.reshape(B, P, T, m).transpose(1, 2).reshape(B * T, P, m)), 1)
        else:
            # same values without the cat passes: the sum is written straight behind the class token, and
            # 'b (p t) -> (b t) p' is ONE strided copy into the buffer that already holds the replicated class tokens
            x1 = torch.empty_like(x)
            x1[:, :1, :] = cls0
            xt = torch.add(x[:, 1:, :], rt, out=x1[:, 1:, :])
            xs_in = torch.empty((B * T, 1 + P, m), dtype=x.dtype, device=x.device)
            xs_in[:, 0, :] = cls0.expand(B, T, m).reshape(B * T, m)
            xs_in.view(B, T, 1 + P, m)[:, :, 1:, :].copy_(xt.reshape(B, P, T, m).transpose(1, 2))
        xs_normed = self.norm1(xs_in)
    # spatial attention per frame, class token replicated into every frame
    rs, metric = self.attn(xs_normed, attn_size)
    rs = self.drop_path(rs)
    cls_new = rs[:, 0, :].reshape(B, T, m).mean(1, keepdim=True)  # class token averaged over frames
    # the second residual cat(cls_new, rearrange(rs, '(b t) p m -> b (p t) m')) stays where the attention left it: the
    # fused merge kernel reads it in the grouped layout (GroupedResidual), nothing is permuted unless a slow path asks
    res = C.GroupedResidual(rs, cls_new, B, T, P)
    # x = cat(cls0, xt) + cat(cls_new, rs); merge per frame; norm2 -- one kernel when the layer merges 16-bit
    # tokens (tome_merge_wavg_regrouped_ln), the reference's steps otherwise
    x, y = C.merge_then_norm_regrouped(
        metric, x1, info, self.norm2,
        lambda z: self.reduction_function(metric, z, info, B, T, P), self.reduction_function is timesformer_merge, T,
        residual=res, fold=C.foldable(self.mlp.fc2, not self.training) if C._plain_mlp(self.mlp) else None)
    return C.mlp_residual(self, self.mlp, x, y, info, drop_path=self.drop_path)


def _attention_forward(self, x, size: torch.Tensor = None):
    """ToMeAttention.forward (timesformer.py:60-83): size bias on the non-class block of the logits only,
    metric = head-mean of the keys without the class token."""
    B, N, Cc = x.shape
    if self.with_qkv:
        q, k, v = self.qkv(x).reshape(B, N, 3, self.num_heads, Cc // self.num_heads).permute(2, 0, 3, 1, 4)
    else:
        q = k = v = x.reshape(B, N, self.num_heads, Cc // self.num_heads).permute(0, 2, 1, 3)
    # the size bias sits on the non-class block of the logits only (timesformer.py:73-74): bias_skip
    drop_p = self.attn_drop.p if self.training else 0.0
    info = getattr(self, "_tome_info", None)
    metric = HeadMeanKeys(k[:, :, 1:, :])  # k.mean(1)[:, 1:, :] averaged inside the matching kernel
    ready = C.keys_ready(metric.keys, info, capture_only=True)  # the keys exist behind the qkv GEMM
    out = C.attention(q, k, v, size, self.scale, drop_p, bias_skip=True)
    C.match_beside(metric, ready, info)  # inside a graph capture: the matching on the side stream, beside the attention
    if self.with_qkv:
        out = self.proj_drop(self.proj(out))
    return out, metric


def _regroup(x, B, T, P):
    """'b (p t) m -> (b t) p m' on the tokens after the class token."""
    return x[:, 1:, :].reshape(B, P, T, -1).transpose(1, 2).reshape(B * T, P, -1)


def _ungroup(cls, y, B, T):
    """'(b t) p m -> b (p t) m' and the class token back in front."""
    P2 = y.size(1)
    return torch.cat((cls, y.reshape(B, T, P2, -1).transpose(1, 2).reshape(B, P2 * T, -1)), dim=1)


def timesformer_merge(metric, x, _tome_info, B, T, num_spatial_tokens):
    r = _tome_info["r"].pop(0)
    if r > 0:
        # 'b (p t) m -> (b t) p m', merge, '(b t) p m -> b (p t) m', cls back in front: done by the kernel's
        # addressing (tome_merge_wavg_regrouped), not by permuted copies
        x = C.reduce_merge_regrouped(metric, x, _tome_info, r, T)
    return x


def timesformer_drop(metric, x, _tome_info, B, T, num_spatial_tokens):
    r = _tome_info["r"].pop(0)
    if r > 0:
        x = C.reduce_drop_regrouped(metric, x, _tome_info, r, T)  # groups addressed in place, no permuted copies
    return x


def timesformer_hybrid(metric, x, _tome_info, B, T, num_spatial_tokens):
    r = _tome_info["r"].pop(0)
    if r > 0:
        x = C.reduce_merge_regrouped(metric, x, _tome_info, r, T, hybrid=True)
    return x


def apply_duplicate_patch(model, layer_to_duplicate, quantity):
    """timesformer.py:170-172: the same block object is visited `quantity` times."""
    for i in range(layer_to_duplicate + 1, layer_to_duplicate + quantity):
        model.model.blocks.insert(index=i, module=model.model.blocks[layer_to_duplicate])


def _is_block(m) -> bool:
    return all(hasattr(m, a) for a in ("attn", "mlp", "norm1", "norm2", "attention_type"))


def apply_patch(model_wrapper, trace_source: bool = False, prop_attn: bool = True, mode: str = "merge",
                head_aggregation: str = "mean", threshold: float = 0.0, verbose: bool = False):
    model = model_wrapper.model
    C.wrap_model_forward(model_wrapper, lambda w: w.model.blocks)
    model_wrapper.r = 0
    info = C.new_tome_info(trace_source, prop_attn, mode, head_aggregation, threshold, verbose, class_token=False)
    del info["head_aggregation"]  # the reference's TimeSformer dict has no such key (timesformer.py:194-205)
    model_wrapper._tome_info = info
    reduction_function = C.pick_reduction(mode, timesformer_merge, timesformer_drop, timesformer_hybrid)
    for module in model.modules():
        if _is_block(module):
            C.swizzle(module, "ToMeBlock", {"forward": _block_forward})
            module._tome_info = model_wrapper._tome_info
            module.reduction_function = reduction_function
            C.swizzle(module.attn, "ToMeAttention", {"forward": _attention_forward})
            module.attn._tome_info = model_wrapper._tome_info  # (the layer's r: its matching may start beside its attention)
    if getattr(model, "attention_type", "divided_space_time") == "divided_space_time":
        # the first LayerNorm of a divided space-time block is temporal_norm1: the previous block's last residual
        # add hands it over fused (tome_add_layernorm)
        C.link_next_norms(model.blocks, "temporal_norm1", skip_first=True)
