"""ToMe patch for ViViT (reference: tome/patch/vivit.py).  One joint sequence with a class token
(class_token=True: token 0 is never merged and unm_idx stays position-sorted).  apply_patch takes the
wrapper whose ``.vivit`` is the HF-style model (``.vivit.encoder.layer``)."""
from __future__ import annotations

import copy
import math
import os

import torch
import torch.nn.functional as F

from . import _common as C
from .. import _abi
from ..merge import HeadMeanKeys

_FUSE_QKV = os.environ.get("TOME_VIVIT_QKV", "1") != "0"  # 0 = three separate query / key / value GEMMs (measurement switch)


def _layer_forward(self, hidden_states, head_mask=None, output_attentions=False):
    """ToMeVivitLayer.forward (vivit.py:18-47)."""
    info = self._tome_info
    attn_size = info["size"] if info["prop_attn"] else None
    outs = self.attention(C.first_norm(self, hidden_states, info, self.layernorm_before), attn_size,
                          info["head_aggregation"], head_mask, output_attentions=output_attentions)
    attention_output, metric, rest = outs[0], outs[1], outs[2:]
    # first residual, merge, layernorm_after -- one kernel when the layer merges 16-bit tokens
    # (tome_merge_wavg_ln with the attention output as addend), the three steps of the reference otherwise
    # VivitOutput = dense -> dropout -> + hidden_states: in eval that is one GEMM accumulating onto hidden_states
    # (finish_linear; its bias folded in by the merge kernel), the next layer's layernorm_before handed over
    eval_mode = not self.training
    hidden_states, normed = C.merge_then_norm(metric, hidden_states, info, self.layernorm_after,
                                              self.reduction_function, vivit_merge, residual=attention_output,
                                              fold=C.foldable(self.output.dense, eval_mode))
    layer_output = self.intermediate(normed)
    if eval_mode:
        layer_output = C.finish_linear(self, hidden_states, layer_output, self.output.dense, info)
    else:
        layer_output = C.finish_block(self, hidden_states, self.output.dropout(self.output.dense(layer_output)), info)
    return (layer_output,) + rest


def _duplicate_layer_forward(self, hidden_states, head_mask=None, output_attentions=False):
    """ToMeDuplicateVivitLayer.forward (vivit.py:50-66): attend for the metric only, then merge."""
    info = self._tome_info
    attn_size = info["size"] if info["prop_attn"] else None
    outs = self.attention(self.layernorm_before(hidden_states), attn_size, info["head_aggregation"], head_mask,
                          output_attentions=output_attentions)
    return [self.reduction_function(outs[1], hidden_states, info)]


def _attention_forward(self, hidden_states, size=None, head_aggregation="mean", head_mask=None,
                       output_attentions=False):
    """ToMeVivitAttention.forward (vivit.py:69-83)."""
    self_outputs = self.attention(hidden_states, size, head_aggregation, head_mask, output_attentions)
    attention_output = self.output(self_outputs[0], hidden_states)
    return (attention_output, self_outputs[1]) + self_outputs[2:]


def _self_attention_forward(self, hidden_states, size=None, head_aggregation="mean", head_mask=None,
                            output_attentions=False):
    """ToMeVivitSelfAttention.forward (vivit.py:86-130): log(size) bias on the keys, metric from the keys
    (class-token row included)."""
    B, N, _ = hidden_states.shape
    H, hd = self.num_attention_heads, self.attention_head_size

    def heads(t):
        return t.view(B, N, H, hd).permute(0, 2, 1, 3)

    qkv_w = _fused_qkv(self, hidden_states)
    if qkv_w is not None:
        # query / key / value as ONE projection (the three weight matrices side by side, cached per module and
        # rebuilt when any of them changes): one [B*N, C] x [C, 3C] GEMM instead of three [C, C] ones; q, k, v stay
        # strided views of its output, which the attention and matching kernels read in place
        q, k, v = torch.nn.functional.linear(hidden_states, qkv_w[0], qkv_w[1]).view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    else:
        q, k, v = heads(self.query(hidden_states)), heads(self.key(hidden_states)), heads(self.value(hidden_states))
    probs = None
    info = getattr(self, "_tome_info", None)
    ready = C.keys_ready(k, info) if head_aggregation == "mean" else None  # the keys exist behind the projection(s)
    if output_attentions or head_mask is not None:
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
        if size is not None:
            scores = scores + _abi.log_of_size(size)[:, None, None, :, 0].to(q.dtype)
        probs = self.dropout(F.softmax(scores, dim=-1))
        if head_mask is not None:
            probs = probs * head_mask
        ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, N, H * hd)
    else:
        drop_p = self.dropout.p if self.training else 0.0
        ctx = C.attention(q, k, v, size, 1.0 / math.sqrt(hd), drop_p)
    if head_aggregation == "mean":
        metric = HeadMeanKeys(k)  # k.mean(1), averaged inside the matching kernel when the layer merges
        C.match_beside(metric, ready, info)  # ... on the side stream, beside the attention and the output projection
    elif head_aggregation == "concat":
        metric = k.transpose(1, 2).reshape(B, N, -1)
    else:
        raise ValueError(f"head_aggregation {head_aggregation!r}")
    return (ctx, metric, probs) if output_attentions else (ctx, metric)


def _fused_qkv(self, hidden_states):
    """(weight [3C, C], bias [3C] or None) of the module's query / key / value projections side by side, or None when
    the three are not plain unhooked nn.Linear layers of one shape in inference on a 16-bit device tensor."""
    lins = (self.query, self.key, self.value)
    if not (_FUSE_QKV and hidden_states.is_cuda and hidden_states.dtype in (torch.bfloat16, torch.float16)
            and not self.training and not torch.is_grad_enabled()
            and all(C._stock_module(m, torch.nn.Linear) for m in lins)):
        return None
    ws = [m.weight for m in lins]
    bs = [m.bias for m in lins]
    if any(w.shape != ws[0].shape or w.dtype != hidden_states.dtype or w.device != hidden_states.device for w in ws) \
            or len({b is None for b in bs}) != 1:
        return None
    key = tuple((t.data_ptr(), t._version) for t in ws + [b for b in bs if b is not None])
    cached = self.__dict__.get("_tome_qkv")
    if cached is None or cached[0] != key:
        w = torch.cat([w.detach() for w in ws], dim=0)
        b = None if bs[0] is None else torch.cat([b.detach() for b in bs], dim=0)
        cached = (key, w, b)
        self.__dict__["_tome_qkv"] = cached  # (not a parameter, not in the state_dict)
    return cached[1], cached[2]


def vivit_merge(metric, x, _tome_info):
    r = _tome_info["r"].pop(0)
    return C.reduce_merge(metric, x, _tome_info, r) if r > 0 else x


def vivit_drop(metric, x, _tome_info):
    r = _tome_info["r"].pop(0)
    return C.reduce_drop(metric, x, _tome_info, r) if r > 0 else x


def vivit_hybrid(metric, x, _tome_info):
    r = _tome_info["r"].pop(0)
    return C.reduce_hybrid(metric, x, _tome_info, r) if r > 0 else x


def _is_layer(m) -> bool:
    return all(hasattr(m, a) for a in ("attention", "intermediate", "output", "layernorm_before", "layernorm_after"))


def _is_attention(m) -> bool:
    return hasattr(m, "attention") and hasattr(m, "output") and hasattr(m.attention, "query")


def _is_self_attention(m) -> bool:
    return all(hasattr(m, a) for a in ("query", "key", "value", "num_attention_heads", "attention_head_size"))


def apply_duplicate_patch(model, layer_to_duplicate, quantity):
    """vivit.py:207-211: insert deep copies that only attend + merge."""
    for i in range(layer_to_duplicate, layer_to_duplicate + quantity - 1):
        model.vivit.encoder.layer.insert(index=i, module=copy.deepcopy(model.vivit.encoder.layer[i]))
        C.swizzle(model.vivit.encoder.layer[i], "ToMeDuplicateVivitLayer", {"forward": _duplicate_layer_forward})
    cfg = getattr(model.vivit, "config", None)
    if cfg is not None and hasattr(cfg, "num_hidden_layers"):
        cfg.num_hidden_layers += quantity  # vivit.py:209 (an HF encoder sizes head_mask from it), +quantity as there


def apply_patch(model_wrapper, trace_source: bool = False, prop_attn: bool = True, mode: str = "merge",
                head_aggregation: str = "mean", threshold: float = 0.0, verbose: bool = False):
    model = model_wrapper.vivit
    C.wrap_model_forward(model_wrapper, lambda w: w.vivit.encoder.layer)
    model_wrapper.r = 0
    model_wrapper._tome_info = C.new_tome_info(trace_source, prop_attn, mode, head_aggregation, threshold, verbose,
                                               class_token=model.embeddings.cls_token is not None)
    reduction_function = C.pick_reduction(mode, vivit_merge, vivit_drop, vivit_hybrid)
    for module in model.modules():
        if C.has_tag(module, "ToMeDuplicateVivitLayer"):
            module._tome_info = model_wrapper._tome_info
            module.reduction_function = reduction_function
        elif _is_layer(module):
            C.swizzle(module, "ToMeVivitLayer", {"forward": _layer_forward})
            module._tome_info = model_wrapper._tome_info
            module.reduction_function = reduction_function
        elif _is_attention(module):
            C.swizzle(module, "ToMeVivitAttention", {"forward": _attention_forward})
        elif _is_self_attention(module):
            C.swizzle(module, "ToMeVivitSelfAttention", {"forward": _self_attention_forward})
            module._tome_info = model_wrapper._tome_info  # (the layer's r: its matching starts beside its attention)
    C.link_next_norms(model.encoder.layer, "layernorm_before", tag="ToMeVivitLayer")
