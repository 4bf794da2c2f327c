#!/usr/bin/env python3
"""The SHORT-key attention shapes in isolation (HIP events): TimeSformer's spatial attention (B*8 sequences of 1 + p
tokens, bias_skip form) and Motionformer's per-frame segments (every query of a clip against the keys of one frame at a
time), resident-K/V kernel (tome_attn_resident.h) vs the streaming kernels (TOME_ATTN_RESIDENT=0), same process.
    python tools/attn_short_bench.py"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

from tome import _abi  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(f, n=20):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
rows = []
for (B, H, N) in ((512, 12, 197), (512, 12, 165), (512, 12, 101), (1024, 12, 197), (64, 12, 197)):
    qkv = torch.randn(B, N, 3, H, 64, device=dev).bfloat16()
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    size = torch.randint(1, 9, (B, N - 1, 1), device=dev).float()
    fl = 4.0 * B * H * N * N * 64
    by = 4.0 * B * H * N * 128
    for label, sz, skip in (("plain", None, False), ("size bias (TimeSformer form)", size, True)):
        res = {}
        for env in ("1", "0"):
            os.environ["TOME_ATTN_RESIDENT"] = env
            with torch.no_grad():
                res[env] = timeit(lambda: _abi.prop_attention(q, k, v, sz, 0.125, bias_skip=skip))
        print(f"spatial {B}x{H}x{N} {label}: resident {res['1']:7.1f} us ({fl / res['1'] / 1e6:5.0f} TFLOP/s, "
              f"{by / res['1'] / 1e6:4.2f} TB/s) | streaming {res['0']:7.1f} us ({fl / res['0'] / 1e6:5.0f} TFLOP/s)", flush=True)
# Motionformer: [B, 1 + S*F, ...] queries against F segments of S keys (tome/patch/motionformer.py:98-121)
for (B, H, S, F) in ((64, 12, 196, 8), (64, 12, 148, 8), (8, 12, 196, 8)):
    N = 1 + S * F
    qkv = torch.randn(B, N, 3, H, 64, device=dev).bfloat16()
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    size = torch.randint(1, 9, (B, S * F), device=dev).float().log()
    fl = 4.0 * B * H * (N - 1) * S * F * 64
    for label, sz in (("plain", None), ("size bias", size)):
        res = {}
        for env in ("1", "0"):
            os.environ["TOME_ATTN_RESIDENT"] = env
            with torch.no_grad():
                res[env] = timeit(lambda: _abi.prop_attention_segments(q[:, :, 1:], k[:, :, 1:], v[:, :, 1:], F, 0.125, sz), n=8)
        print(f"segments {B}x{H}x({S}x{F}) {label}: resident {res['1']:7.1f} us ({fl / res['1'] / 1e6:5.0f} TFLOP/s) | "
              f"streaming {res['0']:7.1f} us ({fl / res['0'] / 1e6:5.0f} TFLOP/s)", flush=True)
