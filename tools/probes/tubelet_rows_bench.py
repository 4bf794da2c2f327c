#!/usr/bin/env python3
"""tome_tubelet_rows against the framework's permute-copy on the four hosts' clip shapes (bf16)."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

from tome import _abi  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, rep=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rep):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / rep * 1e3


for name, B, T, kt, view in (("videomae 384 x 16 frames", 384, 16, 2, "plain"), ("videomae 8 x 16 frames", 8, 16, 2, "plain"),
                             ("timesformer 384 x 8 frames (per frame)", 384, 8, 1, "plain"),
                             ("vivit 64 x 32 frames [B,T,C,H,W]", 64, 32, 2, "btchw")):
    if view == "btchw":
        x = torch.rand(B, T, 3, 224, 224, device=dev).bfloat16().permute(0, 2, 1, 3, 4)
    else:
        x = torch.rand(B, 3, T, 224, 224, device=dev).bfloat16()
    nt = T // kt

    def by_views():
        return x.reshape(B, 3, nt, kt, 14, 16, 14, 16).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, nt * 196, 3 * kt * 256)

    t_k = timed(lambda: _abi.tubelet_rows(x, kt, 16, 16))
    t_v = timed(by_views)
    gb = 2 * x.numel() * 2 / 1e9
    print(f"{name:42s}: kernel {t_k:8.1f} us = {gb / t_k * 1e3:5.2f} TB/s   framework permute-copy {t_v:8.1f} us = "
          f"{gb / t_v * 1e3:5.2f} TB/s", flush=True)
    del x
