#!/usr/bin/env python3
"""Workload for a kernel trace of the matching alone: tome_match_keys on random bf16 keys of the benchmark's first
layer shape (batch x 12 heads x 1568 tokens x 64), a few calls.
    rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/match_prof.py [batch]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

from tome import _abi  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 384
dev = torch.device("cuda", 0)
torch.manual_seed(0)
qkv = torch.randn(batch, 1568, 3, 12, 64, device=dev).bfloat16()
keys = qkv.permute(2, 0, 3, 1, 4)[1]
for _ in range(12):
    _abi.match_keys(keys, 16)
torch.cuda.synchronize()
