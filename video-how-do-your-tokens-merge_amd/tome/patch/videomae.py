"""ToMe patch for VideoMAE (reference: tome/patch/videomae.py).  Joint space-time tokens, no class
token; the merge sits between the attention residual and the MLP of every block.

apply_patch(model_wrapper, ...) takes the wrapper whose ``.model`` is the ViT (``.model.blocks``), mutates
it in place and returns None, exactly like the reference; afterwards set ``model_wrapper.r``.
"""
from __future__ import annotations

import copy

import torch
import torch.nn.functional as F

from . import _common as C
from ..merge import HeadMeanKeys


def _block_forward(self, x):
    """ToMeBlock.forward (videomae.py:14-30)."""
    info = self._tome_info
    attn_size = info["size"] if info["prop_attn"] else None
    attn, metric = self.attn(C.first_norm(self, x, info, self.norm1), attn_size, info["head_aggregation"])
    if self.gamma_1 is not None:
        attn = self.gamma_1 * attn
    # x = x + attn; x = merge(x); y = norm2(x) -- one kernel when the layer merges 16-bit tokens
    # (tome_merge_wavg_ln with the residual as addend), the three steps of the reference otherwise
    # (fold: x comes back with fc2's bias in it when `x + mlp(...)` below can be fc2's GEMM accumulating onto x)
    fold = C.foldable(self.mlp.fc2, not self.training) if (self.gamma_2 is None and C._plain_mlp(self.mlp)) else None
    x, y = C.merge_then_norm(metric, x, info, self.norm2, self.reduction_function, videomae_merge,
                             residual=self.drop_path(attn), fold=fold)
    # x + mlp(...), and the next block's norm1 of it handed over when that is possible
    return C.mlp_residual(self, self.mlp, x, y, info, scale=self.gamma_2, drop_path=self.drop_path)


def _duplicate_block_forward(self, x):
    """ToMeDuplicateBlock.forward (videomae.py:33-44): attend only to obtain the metric, then merge."""
    info = self._tome_info
    attn_size = info["size"] if info["prop_attn"] else None
    _, metric = self.attn(self.norm1(x), attn_size, info["head_aggregation"])
    return self.reduction_function(metric, x, info)


def _attention_forward(self, x, size: torch.Tensor = None, head_aggregation: str = "mean"):
    """ToMeAttention.forward (videomae.py:48-77): proportional attention + the key metric."""
    B, N, _ = x.shape
    bias = None
    if self.q_bias is not None:
        bias = torch.cat((self.q_bias, torch.zeros_like(self.v_bias, requires_grad=False), self.v_bias))
    qkv = F.linear(x, self.qkv.weight, bias).reshape(B, N, 3, self.num_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    # softmax(q*scale @ k^T + log(size)) @ v: q, k, v are read in place from the qkv buffer
    drop_p = self.attn_drop.p if self.training else 0.0
    info = getattr(self, "_tome_info", None)
    ready = C.keys_ready(k, info) if head_aggregation == "mean" else None  # the keys exist behind the qkv GEMM
    out = C.attention(q, k, v, size, self.scale, drop_p)
    if head_aggregation == "mean":
        metric = HeadMeanKeys(k)  # k.mean(1), averaged inside the matching kernel when the layer merges
        C.match_beside(metric, ready, info)  # ... on the side stream, beside the attention and the projection
    elif head_aggregation == "concat":
        metric = k.transpose(1, 2).reshape(B, N, -1)
    else:
        raise ValueError(f"head_aggregation {head_aggregation!r}")
    return self.proj_drop(self.proj(out)), metric


def videomae_merge(metric, x, _tome_info):
    r = _tome_info["r"].pop(0)
    return C.reduce_merge(metric, x, _tome_info, r) if r > 0 else x


def videomae_drop(metric, x, _tome_info):
    r = _tome_info["r"].pop(0)
    return C.reduce_drop(metric, x, _tome_info, r) if r > 0 else x


def videomae_hybrid(metric, x, _tome_info):
    r = _tome_info["r"].pop(0)
    return C.reduce_hybrid(metric, x, _tome_info, r) if r > 0 else x


def _is_block(m) -> bool:
    return all(hasattr(m, a) for a in ("attn", "mlp", "norm1", "norm2", "gamma_1"))


def _is_attention(m) -> bool:
    return all(hasattr(m, a) for a in ("qkv", "proj", "num_heads", "q_bias"))


def apply_duplicate_patch(model, layer_to_duplicate, quantity):
    """videomae.py:154-157: insert deep copies that only attend + merge."""
    for i in range(layer_to_duplicate, layer_to_duplicate + quantity - 1):
        model.model.blocks.insert(index=i, module=copy.deepcopy(model.model.blocks[i]))
        C.swizzle(model.model.blocks[i], "ToMeDuplicateBlock", {"forward": _duplicate_block_forward})


def apply_patch(model_wrapper, trace_source: bool = False, prop_attn: bool = False, mode: str = "merge",
                head_aggregation: str = "mean", threshold: float = 0.0, verbose: bool = False):
    model = model_wrapper.model
    C.wrap_model_forward(model_wrapper, lambda w: w.model.blocks)
    model_wrapper.r = 0
    model_wrapper._tome_info = C.new_tome_info(trace_source, prop_attn, mode, head_aggregation, threshold, verbose,
                                               class_token=False)
    reduction_function = C.pick_reduction(mode, videomae_merge, videomae_drop, videomae_hybrid)
    for module in model.modules():
        if C.has_tag(module, "ToMeDuplicateBlock"):
            module._tome_info = model_wrapper._tome_info
            module.reduction_function = reduction_function
        elif _is_block(module):
            C.swizzle(module, "ToMeBlock", {"forward": _block_forward})
            module._tome_info = model_wrapper._tome_info
            module.reduction_function = reduction_function
        elif _is_attention(module):
            C.swizzle(module, "ToMeAttention", {"forward": _attention_forward})
            module._tome_info = model_wrapper._tome_info  # (the layer's r: its matching starts beside its attention)
    C.link_next_norms(model.blocks, "norm1")
