"""TimeSformer host (architecture of slowfast/models/timesformer.py:57-319 in the reference: per-frame
Conv2d patches, class token, learned space + time embeddings, divided space-time blocks).  Token layout
after the class token is patch-major: index 1 + p*T + t.  Parameter names match the reference."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ._patchify import frame_patch_tokens, patch_tokens, tubelet_tokens  # noqa: F401
from tome import _abi

_SHORT_KERNEL = os.environ.get("TOME_SHORT_ATTN", "1") != "0"  # 0 = the framework's attention for the temporal stage


class Mlp(nn.Module):
    def __init__(self, dim, hidden, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0, with_qkv=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.with_qkv = with_qkv
        if with_qkv:
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.proj = nn.Linear(dim, dim)
            self.proj_drop = nn.Dropout(proj_drop)
        self.attn_drop = nn.Dropout(attn_drop)

    def forward(self, x):
        B, N, C = x.shape
        if self.with_qkv:
            q, k, v = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        else:
            q = k = v = x.reshape(B, N, self.num_heads, C // self.num_heads).permute(0, 2, 1, 3)
        if N <= 8 and _SHORT_KERNEL and q.is_cuda and not (self.training and self.attn_drop.p > 0.0) \
                and _abi.short_attention_ok(q, k, v):
            # the temporal attention: thousands of sequences of T <= 8 tokens -- one streaming pass over q, k, v
            # (tome_short_attention: 5 TB/s; the framework's fused attention + the head transpose run at 2.8)
            x = _abi.short_attention(q, k, v, self.scale, checked=True)
        else:
            x = F.scaled_dot_product_attention(q, k, v, scale=self.scale).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x)) if self.with_qkv else x


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, attention_type="divided_space_time", eps=1e-6):
        super().__init__()
        assert attention_type in ("divided_space_time", "space_only", "joint_space_time")
        self.attention_type = attention_type
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias)
        if attention_type == "divided_space_time":
            self.temporal_norm1 = nn.LayerNorm(dim, eps=eps)
            self.temporal_attn = Attention(dim, num_heads, qkv_bias)
            self.temporal_fc = nn.Linear(dim, dim)
        self.drop_path = nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, B, T, W):
        if self.attention_type != "divided_space_time":
            x = x + self.attn(self.norm1(x))
            return x + self.mlp(self.norm2(x))
        P = (x.size(1) - 1) // T
        m = x.size(2)
        xt = x[:, 1:, :]
        rt = self.temporal_attn(self.temporal_norm1(xt.reshape(B * P, T, m))).reshape(B, P * T, m)
        xt = xt + self.temporal_fc(rt)
        cls0 = x[:, :1, :]
        cls = cls0.expand(B, T, m).reshape(B * T, 1, m)
        xs = xt.reshape(B, P, T, m).transpose(1, 2).reshape(B * T, P, m)
        rs = self.attn(self.norm1(torch.cat((cls, xs), 1)))
        cls_new = rs[:, 0, :].reshape(B, T, m).mean(1, keepdim=True)
        rs = rs[:, 1:, :].reshape(B, T, P, m).transpose(1, 2).reshape(B, P * T, m)
        x = torch.cat((cls0, xt), 1) + torch.cat((cls_new, rs), 1)
        return x + self.mlp(self.norm2(x))


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        B, C, T, H, W = x.shape
        x = frame_patch_tokens(self.proj, x)
        return x, T, W // self.proj.kernel_size[1]


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=400, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=True, num_frames=8, attention_type="divided_space_time"):
        super().__init__()
        self.attention_type = attention_type
        self.depth = depth
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        if attention_type != "space_only":
            self.time_embed = nn.Parameter(torch.zeros(1, num_frames, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, attention_type)
                                     for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init)
        if attention_type == "divided_space_time":
            for i, blk in enumerate(self.blocks):  # later blocks start with an inert temporal branch
                if i > 0:
                    nn.init.zeros_(blk.temporal_fc.weight)
                    nn.init.zeros_(blk.temporal_fc.bias)

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
        B = x.shape[0]
        x, T, W = self.patch_embed(x)                       # [(B T), n, m]
        if self.attention_type != "space_only" and not (torch.is_grad_enabled() and x.requires_grad):
            # cat(cls, x) + pos_embed, 'b t n -> b n t', + time_embed, cat(cls, x) of the reference -- the same sums
            # in the same order (patch + pos, rounded, + time; cls + pos[0]) written straight into the token sequence
            # [B, 1 + n*T, m]: two passes over the tokens instead of six
            n, m = x.size(1), x.size(2)
            seq = torch.empty((B, 1 + n * T, m), dtype=x.dtype, device=x.device)
            body = seq[:, 1:].view(B, n, T, m)
            torch.add(x.view(B, T, n, m).transpose(1, 2), self.pos_embed[:, 1:].view(1, n, 1, m), out=body)
            body.add_(self.time_embed.view(1, 1, T, m))
            seq[:, 0] = self.cls_token[:, 0] + self.pos_embed[:, 0]
            x = seq
        else:
            x = torch.cat((self.cls_token.expand(x.size(0), -1, -1), x), dim=1) + self.pos_embed
            if self.attention_type != "space_only":
                cls = x[:B, 0, :].unsqueeze(1)
                n = x.size(1) - 1
                x = x[:, 1:].reshape(B, T, n, -1).transpose(1, 2).reshape(B * n, T, -1) + self.time_embed
                x = torch.cat((cls, x.reshape(B, n * T, -1)), dim=1)
        for blk in self.blocks:
            x = blk(x, B, T, W)
        if self.attention_type == "space_only":
            x = x.reshape(B, T, x.size(1), -1).mean(1)
        return self.norm(x)[:, 0]

    def forward(self, x):
        return self.head(self.forward_features(x[0]))


class TimeSformer(nn.Module):
    """Wrapper with ``.model`` like slowfast's TimeSformer (timesformer.py:332-350), without the
    pretrained-weight download."""

    def __init__(self, num_frames=8, num_classes=400, attention_type="divided_space_time", **vit_kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.attention_type = attention_type
        self.model = VisionTransformer(num_classes=num_classes, num_frames=num_frames, attention_type=attention_type,
                                       **vit_kwargs)

    def forward(self, x):
        return self.model(x)


def timesformer_base(num_frames=8, **kw) -> TimeSformer:
    return TimeSformer(num_frames=num_frames, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                       qkv_bias=True, **kw)
