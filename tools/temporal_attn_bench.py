#!/usr/bin/env python3
"""TimeSformer's temporal attention in isolation: 64 x P sequences of 8 tokens (12 heads x 64) as strided views of
one qkv buffer -- PyTorch-ROCm's fused attention + the head transpose against tome_short_attention (HIP events).
    python tools/temporal_attn_bench.py"""
import torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for P in (196, 148, 100, 52, 20):
    B = 64 * P
    qkv = torch.randn(B, 8, 3, 12, 64, device=dev).bfloat16()
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    with torch.no_grad():
        t = timeit(lambda: F.scaled_dot_product_attention(q, k, v, scale=0.125).transpose(1, 2).reshape(B, 8, 768))
    byt = B * 8 * 768 * 2 * 4
    print(f"P={P}: {B} sequences, {t:.1f} us, {byt/t/1e6:.2f} TB/s", flush=True)
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
from tome import _abi
for P in (196, 148, 100, 52, 20):
    B = 64 * P
    qkv = torch.randn(B, 8, 3, 12, 64, device=dev).bfloat16()
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    t = timeit(lambda: _abi.short_attention(q, k, v, 0.125))
    byt = B * 8 * 768 * 2 * 4
    print(f"ours P={P}: {B} sequences, {t:.1f} us, {byt/t/1e6:.2f} TB/s", flush=True)
