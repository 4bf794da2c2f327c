"""Non-overlapping patch / tubelet embeddings as ONE matrix product.

Every model of the reference embeds its clip with a convolution whose stride equals its kernel
(VideoMAE `PatchEmbed.proj` Conv3d 2x16x16, slowfast/models/videomae_video_model_builder.py:137-166;
TimeSformer per-frame Conv2d 16x16; Motionformer `PatchEmbed3D`; ViViT tubelet Conv3d): the patches do not
overlap, so the convolution is `patches[B*N, C*kt*kh*kw] @ weight[out, C*kt*kh*kw]^T + bias`.  On MI355X the
library convolution for this shape is the slowest kernel of the whole forward (5.5 ms of a 35 ms VideoMAE-B step
at batch 64, plus layout conversions); the regrouping copy + hipBLASLt GEMM takes well under 1 ms.  Same weights,
same parameter names, same token order (t', h', w' row-major = `conv(x).flatten(2).transpose(1, 2)`)."""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from tome import _abi

_ROWS_KERNEL = os.environ.get("TOME_ROWS_KERNEL", "1") != "0"  # 0 = the framework's permute-copy in front of the GEMM


def _plain(conv) -> bool:
    k = tuple(conv.kernel_size)
    return (tuple(conv.stride) == k and all(p == 0 for p in conv.padding) and all(d == 1 for d in conv.dilation)
            and conv.groups == 1 and isinstance(conv.padding, tuple))


def _rows(x: torch.Tensor, kt: int, kh: int, kw: int) -> torch.Tensor:
    """[B, T'*H'*W', C*kt*kh*kw]: the regrouped clip.  On the device one 16-byte move kernel that reads any view with
    unit stride along W where it lies (tome_tubelet_rows: 0.3 ms for 384 VideoMAE clips where the framework's
    permute-copy takes 1.8); the framework's reshape / permute otherwise."""
    B, C, T, H, W = x.shape
    if _ROWS_KERNEL and _abi.tubelet_rows_ok(x, kt, kh, kw):
        return _abi.tubelet_rows(x, kt, kh, kw)
    nt, nh, nw = T // kt, H // kh, W // kw
    return x.reshape(B, C, nt, kt, nh, kh, nw, kw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, nt * nh * nw, C * kt * kh * kw)


def tubelet_tokens(conv: torch.nn.Conv3d, x: torch.Tensor) -> torch.Tensor:
    """x [B, C, T, H, W] -> tokens [B, T'*H'*W', out_channels], equal to conv(x).flatten(2).transpose(1, 2)."""
    kt, kh, kw = conv.kernel_size
    B, C, T, H, W = x.shape
    if not _plain(conv) or T % kt or H % kh or W % kw:
        return conv(x).flatten(2).transpose(1, 2)
    return F.linear(_rows(x, kt, kh, kw), conv.weight.reshape(conv.out_channels, -1), conv.bias)


def frame_patch_tokens(conv: torch.nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """x [B, C, T, H, W] -> per-frame tokens [B*T, H'*W', out_channels], equal to
    conv(rearrange(x, 'b c t h w -> (b t) c h w')).flatten(2).transpose(1, 2): a tubelet of one frame, the clip read
    where it lies (no '(b t) c h w' copy in front)."""
    kh, kw = conv.kernel_size
    B, C, T, H, W = x.shape
    if not _plain(conv) or H % kh or W % kw:
        return conv(x.transpose(1, 2).reshape(B * T, C, H, W)).flatten(2).transpose(1, 2)
    rows = _rows(x, 1, kh, kw)  # [B, T*H'*W', C*kh*kw], tokens (t, h', w') row-major
    return F.linear(rows.view(B * T, (H // kh) * (W // kw), C * kh * kw), conv.weight.reshape(conv.out_channels, -1), conv.bias)


def patch_tokens(conv: torch.nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """x [n, C, H, W] -> tokens [n, H'*W', out_channels], equal to conv(x).flatten(2).transpose(1, 2)."""
    kh, kw = conv.kernel_size
    n, C, H, W = x.shape
    if not _plain(conv) or H % kh or W % kw:
        return conv(x).flatten(2).transpose(1, 2)
    nh, nw = H // kh, W // kw
    p = x.reshape(n, C, nh, kh, nw, kw).permute(0, 2, 4, 1, 3, 5).reshape(n, nh * nw, C * kh * kw)
    return F.linear(p, conv.weight.reshape(conv.out_channels, -1), conv.bias)
