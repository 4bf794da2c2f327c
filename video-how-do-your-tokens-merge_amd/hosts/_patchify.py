"""Non-overlapping patch / tubelet embeddings as ONE matrix product.

Every model of the reference embeds its clip with a convolution whose stride equals its kernel
(VideoMAE `PatchEmbed.proj` Conv3d 2x16x16, slowfast/models/videomae_video_model_builder.py:137-166;
TimeSformer per-frame Conv2d 16x16; Motionformer `PatchEmbed3D`; ViViT tubelet Conv3d): the patches do not
overlap, so the convolution is `patches[B*N, C*kt*kh*kw] @ weight[out, C*kt*kh*kw]^T + bias`.  On MI355X the
library convolution for this shape is the slowest kernel of the whole forward (5.5 ms of a 35 ms VideoMAE-B step
at batch 64, plus layout conversions); the regrouping copy + hipBLASLt GEMM takes well under 1 ms.  Same weights,
same parameter names, same token order (t', h', w' row-major = `conv(x).flatten(2).transpose(1, 2)`)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _plain(conv) -> bool:
    k = tuple(conv.kernel_size)
    return (tuple(conv.stride) == k and all(p == 0 for p in conv.padding) and all(d == 1 for d in conv.dilation)
            and conv.groups == 1 and isinstance(conv.padding, tuple))


def tubelet_tokens(conv: torch.nn.Conv3d, x: torch.Tensor) -> torch.Tensor:
    """x [B, C, T, H, W] -> tokens [B, T'*H'*W', out_channels], equal to conv(x).flatten(2).transpose(1, 2)."""
    kt, kh, kw = conv.kernel_size
    B, C, T, H, W = x.shape
    if not _plain(conv) or T % kt or H % kh or W % kw:
        return conv(x).flatten(2).transpose(1, 2)
    nt, nh, nw = T // kt, H // kh, W // kw
    p = x.reshape(B, C, nt, kt, nh, kh, nw, kw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, nt * nh * nw, C * kt * kh * kw)
    return F.linear(p, conv.weight.reshape(conv.out_channels, -1), conv.bias)


def patch_tokens(conv: torch.nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """x [n, C, H, W] -> tokens [n, H'*W', out_channels], equal to conv(x).flatten(2).transpose(1, 2)."""
    kh, kw = conv.kernel_size
    n, C, H, W = x.shape
    if not _plain(conv) or H % kh or W % kw:
        return conv(x).flatten(2).transpose(1, 2)
    nh, nw = H // kh, W // kw
    p = x.reshape(n, C, nh, kh, nw, kw).permute(0, 2, 4, 1, 3, 5).reshape(n, nh * nw, C * kh * kw)
    return F.linear(p, conv.weight.reshape(conv.out_channels, -1), conv.bias)
