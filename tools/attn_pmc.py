#!/usr/bin/env python3
"""Workload of the attention kernel's counter passes: a few launches of tome_prop_attention (plain and with the
size bias) at the headline shape (128 x 12 heads x 1568 tokens, bf16).  Run under
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -- python3 tools/attn_pmc.py
(tools/attn_pmc.sh does the passes), then tools/attn_pmc_parse.py <dir>... > profiles/<round>_attention_pmc.json"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

from tome import _abi  # noqa: E402

dev = torch.device("cuda", 0)
B, H, N = 128, 12, 1568
torch.manual_seed(0)
qkv = torch.randn(B, N, 3, H, 64, device=dev).bfloat16()
q, k, v = qkv.permute(2, 0, 3, 1, 4)
size = torch.randint(1, 9, (B, N, 1), device=dev).float()
for _ in range(4):
    _abi.prop_attention(q, k, v, None, 0.125)
    _abi.prop_attention(q, k, v, size, 0.125)
torch.cuda.synchronize()
del qkv, q, k, v
# round 4: the resident-K/V kernel on its two shapes -- Motionformer's per-frame segments (64 x 12 heads, 8 segments of
# 196 keys, 1568 queries) and TimeSformer's frames (512 x 12 x 197, bias_skip form)
B, S, F = 64, 196, 8
qkv = torch.randn(B, 1 + S * F, 3, H, 64, device=dev).bfloat16()
q, k, v = qkv.permute(2, 0, 3, 1, 4)
for _ in range(4):
    _abi.prop_attention_segments(q[:, :, 1:], k[:, :, 1:], v[:, :, 1:], F, 0.125)
torch.cuda.synchronize()
del qkv, q, k, v
qkv = torch.randn(512, 197, 3, H, 64, device=dev).bfloat16()
q, k, v = qkv.permute(2, 0, 3, 1, 4)
size = torch.randint(1, 9, (512, 196, 1), device=dev).float()
for _ in range(4):
    _abi.prop_attention(q, k, v, size, 0.125, bias_skip=True)
torch.cuda.synchronize()
