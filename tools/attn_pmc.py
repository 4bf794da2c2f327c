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
