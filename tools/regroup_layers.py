#!/usr/bin/env python3
"""Per-layer time of the regrouped merge kernel (residual add + merge + LayerNorm on TimeSformer's / Motionformer's
'(b t) p' frame groups, tome_merge_wavg_regrouped_ln) along an r schedule: tokens, bytes moved, microseconds, TB/s.
    python tools/regroup_layers.py [batch] [r] [frames] [tokens]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

from tome import _abi  # noqa: E402

batch, r, frames, t = [int(v) for v in (sys.argv[1:5] + ["64", "32", "8", "196"][len(sys.argv) - 1:])]
dev = torch.device("cuda", 0)
EMBED, H = 768, 12
g = torch.Generator(device=dev).manual_seed(11)
n = batch * frames
x = torch.randn(batch, 1 + t * frames, EMBED, device=dev, generator=g).bfloat16()
ln_w = torch.ones(EMBED, device=dev).bfloat16()
ln_b = torch.zeros(EMBED, device=dev).bfloat16()
size = None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tb = tt = 0.0
for layer in range(12):
    re = max(0, min(r, t // 2))
    if re <= 0:
        break
    keys = torch.randn(n, H, t, 64, device=dev, generator=g).bfloat16()
    plan = _abi.match_keys(keys, re, False)
    res = (0.1 * torch.randn(x.shape, device=dev, generator=g)).bfloat16()
    f = lambda: _abi.merge_wavg_regrouped(plan, x, size, frames, has_cls=True, ln=(ln_w, ln_b, 1e-6), addend=res)
    for _ in range(3):
        out = f()
    e0.record()
    for _ in range(10):
        out = f()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 100
    rows_in, rows_out = batch * (1 + t * frames), batch * (1 + (t - re) * frames)
    nbytes = 2 * rows_in * EMBED * 2 + 2 * rows_out * EMBED * 2 + n * t * 2 + n * (t - re) * 2
    tb += nbytes
    tt += us
    print(f"layer {layer:2d}: {t:4d} -> {t - re:4d} tokens per frame group, {nbytes / 1e6:7.1f} MB, {us:7.1f} us, {nbytes / us / 1e6:5.2f} TB/s")
    x, size = out[0], out[2]
    t -= re
print(f"total {tb / 1e6:.1f} MB in {tt:.1f} us: {tb / tt / 1e6:.2f} TB/s = {tb / tt / 1e6 / 8:.3f} of 8 TB/s")
