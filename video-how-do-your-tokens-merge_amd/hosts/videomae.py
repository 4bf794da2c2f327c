"""VideoMAE ViT host (architecture of slowfast/models/videomae_video_model_builder.py:59-304 in the
reference: tubelet Conv3d embedding, fixed sin-cos positions, pre-LN blocks with q/v-only qkv bias, mean
pooling + fc_norm head).  Parameter names match the reference so its checkpoints load with
``load_state_dict``; the ToMe patch (tome.patch.videomae) takes the ``VideoMAE`` wrapper.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ._patchify import patch_tokens, tubelet_tokens  # noqa: F401


def sincos_table(n_position: int, d_hid: int) -> torch.Tensor:
    """Fixed positional table: even channels sin, odd channels cos of pos / 10000^(2*(j//2)/d), evaluated in
    numpy float64 and cast once (bit-identical to the table the reference's checkpoints were trained with)."""
    j = np.arange(d_hid)
    angle = np.arange(n_position, dtype=np.float64)[:, None] / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    angle[:, 0::2] = np.sin(angle[:, 0::2])
    angle[:, 1::2] = np.cos(angle[:, 1::2])
    return torch.tensor(angle, dtype=torch.float).unsqueeze(0)


class Mlp(nn.Module):
    def __init__(self, dim, hidden, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.act(self.fc1(x))))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=12, qkv_bias=True, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim)) if qkv_bias else None
        self.v_bias = nn.Parameter(torch.zeros(dim)) if qkv_bias else None
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, _ = x.shape
        bias = None
        if self.q_bias is not None:
            bias = torch.cat((self.q_bias, torch.zeros_like(self.v_bias), self.v_bias))
        q, k, v = F.linear(x, self.qkv.weight, bias).reshape(B, N, 3, self.num_heads, -1).permute(2, 0, 3, 1, 4)
        out = F.scaled_dot_product_attention(q, k, v, scale=self.scale)
        return self.proj_drop(self.proj(out.transpose(1, 2).reshape(B, N, -1)))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, init_values=0.0, eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.drop_path = nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        if init_values > 0:
            self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
            self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))
        else:
            self.gamma_1, self.gamma_2 = None, None

    def forward(self, x):
        a = self.attn(self.norm1(x))
        x = x + (a if self.gamma_1 is None else self.gamma_1 * a)
        m = self.mlp(self.norm2(x))
        return x + (m if self.gamma_2 is None else self.gamma_2 * m)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, num_frames=16, tubelet_size=2):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2 * (num_frames // tubelet_size)
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=(tubelet_size, patch_size, patch_size),
                              stride=(tubelet_size, patch_size, patch_size))

    def forward(self, x):
        return tubelet_tokens(self.proj, x)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=400, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=True, init_values=0.0, all_frames=16, tubelet_size=2,
                 use_mean_pooling=True, init_scale=0.001):
        super().__init__()
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, all_frames, tubelet_size)
        self.register_buffer("pos_embed", sincos_table(self.patch_embed.num_patches, embed_dim), persistent=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, init_values)
                                     for _ in range(depth)])
        self.norm = nn.Identity() if use_mean_pooling else nn.LayerNorm(embed_dim, eps=1e-6)
        self.fc_norm = nn.LayerNorm(embed_dim, eps=1e-6) if use_mean_pooling else None
        self.head = nn.Linear(embed_dim, num_classes)
        self.apply(self._init)
        nn.init.trunc_normal_(self.head.weight, std=0.02)
        self.head.weight.data.mul_(init_scale)
        self.head.bias.data.mul_(init_scale)

    @staticmethod
    def _init(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    def forward_features(self, x):
        x = x[0]  # slowfast models take a list of pathways
        x = self.patch_embed(x)
        x = x + self.pos_embed.to(x.dtype)
        for blk in self.blocks:
            x = blk(x)
        x = self.norm(x)
        return self.fc_norm(x.mean(1)) if self.fc_norm is not None else x[:, 0]

    def forward(self, x):
        return self.head(self.forward_features(x))


class VideoMAE(nn.Module):
    """Wrapper with ``.model`` like slowfast's VideoMAE (videomae_video_model_builder.py:363-397)."""

    def __init__(self, num_frames=16, num_classes=400, tubelet_size=2, **vit_kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.model = VisionTransformer(num_classes=num_classes, all_frames=num_frames, tubelet_size=tubelet_size,
                                       **vit_kwargs)

    def forward(self, x):
        return self.model(x)


def videomae_base(num_frames=16, **kw) -> VideoMAE:
    """VideoMAE-B: 768 wide, 12 layers, 12 heads -- vit_base_patch16_224 of the reference."""
    return VideoMAE(num_frames=num_frames, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                    qkv_bias=True, **kw)
