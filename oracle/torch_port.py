"""PyTorch-CPU port of the reference's merge path and of the VideoMAE ToMe forward around it.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  This is what ``bench.py`` times as the
``cpu_baseline`` (kind "port"): the same ATen op sequence the reference executes on CPU
(tome/merge.py:51-85,363-368 inside tome/patch/videomae.py:14-100 and
slowfast/models/videomae_video_model_builder.py:272-304), restated here because the reference does not
travel to the GPU box.  tests/test_torch_port.py pins it against the golden vectors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class TorchPlan:
    r: int
    src_idx: torch.Tensor  # [n,r,1] int64
    dst_idx: torch.Tensor  # [n,r,1]
    unm_idx: torch.Tensor  # [n,T1-r,1]
    tokens: int


def match(metric: torch.Tensor, r: int, class_token: bool = False, distill_token: bool = False) -> Optional[TorchPlan]:
    """merge.py:36-73: clamp r, cosine scores between even and odd tokens, row max, descending argsort."""
    tokens = metric.shape[1]
    r = min(r, (tokens - int(class_token) - int(distill_token)) // 2)
    if r <= 0:
        return None
    with torch.no_grad():
        unit = metric / metric.norm(dim=-1, keepdim=True)
        scores = unit[..., ::2, :] @ unit[..., 1::2, :].transpose(-1, -2)
        if class_token:
            scores[..., 0, :] = -math.inf
        if distill_token:
            scores[..., :, 0] = -math.inf
        node_max, node_idx = scores.max(dim=-1)
        order = node_max.argsort(dim=-1, descending=True)[..., None]
        unm_idx, src_idx = order[..., r:, :], order[..., :r, :]
        dst_idx = node_idx[..., None].gather(dim=-2, index=src_idx)
        if class_token:
            unm_idx = unm_idx.sort(dim=1)[0]
    return TorchPlan(r, src_idx, dst_idx, unm_idx, tokens)


def merge(plan: TorchPlan, x: torch.Tensor, mode: str = "mean") -> torch.Tensor:
    """merge.py:75-85 (no distill layout: never enabled in the reference's patches)."""
    even, odd = x[..., ::2, :], x[..., 1::2, :]
    n, t1, c = even.shape
    kept = even.gather(dim=-2, index=plan.unm_idx.expand(n, t1 - plan.r, c))
    moved = even.gather(dim=-2, index=plan.src_idx.expand(n, plan.r, c))
    odd = odd.scatter_reduce(-2, plan.dst_idx.expand(n, plan.r, c), moved, reduce=mode)
    return torch.cat([kept, odd], dim=1)


def merge_wavg(plan: TorchPlan, x: torch.Tensor, size: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """merge.py:355-369."""
    if size is None:
        size = torch.ones_like(x[..., 0, None])
    x = merge(plan, x * size, mode="sum")
    size = merge(plan, size, mode="sum")
    return x / size, size


def schedule(num_layers: int, r) -> List[int]:
    """tome/utils.py:83-108 (int, (r, inflect) or list)."""
    if isinstance(r, list):
        return list(r) + [0] * max(0, num_layers - len(r))
    inflect = 0
    if isinstance(r, tuple):
        r, inflect = r
    lo = int(r * (1.0 - inflect))
    step = ((2 * r - lo) - lo) / (num_layers - 1)
    return [int(lo + step * i) for i in range(num_layers)]


@torch.no_grad()
def videomae_forward(host, clips: torch.Tensor, r, prop_attn: bool = False, trace: Optional[list] = None) -> torch.Tensor:
    """Forward of a ``hosts.videomae.VideoMAE`` (un-patched, on CPU) with ToMe merging, written as the
    reference runs it: explicit q@k^T softmax attention (videomae.py:48-77), metric = k.mean(heads),
    merge between attention and MLP (videomae.py:14-30), mean pooling head.  ``trace`` collects
    (tokens_before, plan) per layer."""
    vit = host.model
    x = vit.patch_embed(clips)
    x = x + vit.pos_embed.to(x.dtype)
    rs = schedule(len(vit.blocks), r)
    size = None
    for blk, r_l in zip(vit.blocks, rs):
        a = blk.attn
        h = blk.norm1(x)
        B, N, _ = h.shape
        bias = None
        if a.q_bias is not None:
            bias = torch.cat((a.q_bias, torch.zeros_like(a.v_bias), a.v_bias))
        qkv = F.linear(h, a.qkv.weight, bias).reshape(B, N, 3, a.num_heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = (q * a.scale) @ k.transpose(-2, -1)
        if prop_attn and size is not None:
            att = att + size.log()[:, None, None, :, 0]
        att = att.softmax(dim=-1)
        out = a.proj((att @ v).transpose(1, 2).reshape(B, N, -1))
        x = x + (out if blk.gamma_1 is None else blk.gamma_1 * out)
        metric = k.mean(1)
        if r_l > 0:
            plan = match(metric, r_l)
            if plan is not None:
                if trace is not None:
                    trace.append((x.shape[1], plan))
                x, size = merge_wavg(plan, x, size)
        m = blk.mlp(blk.norm2(x))
        x = x + (m if blk.gamma_2 is None else blk.gamma_2 * m)
    x = vit.norm(x)
    x = vit.fc_norm(x.mean(1)) if vit.fc_norm is not None else x[:, 0]
    return vit.head(x)
