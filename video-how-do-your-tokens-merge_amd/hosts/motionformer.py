"""Motionformer host (architecture of slowfast/models/motionformer_video_model_builder.py:25-283 and
motionformer_vit_helper.py:146-318 in the reference: 3-D tubelet patches, class token, separate space +
time embeddings, trajectory attention, tanh pre-logits, softmax head in eval mode).  The ToMe patch takes
the model itself (no wrapper).  Unlike the reference the tubelet projection is NOT zero-initialised (it
would make all random-init tokens identical)."""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from ._patchify import patch_tokens, tubelet_tokens  # noqa: F401
from einops import rearrange


def qkv_attn(q, k, v):
    attn = torch.einsum("b i d, b j d -> b i j", q, k).softmax(dim=-1)
    return torch.einsum("b i j, b j d -> b i d", attn, v)


class Mlp(nn.Module):
    def __init__(self, dim, hidden, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class TrajectoryAttention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0, use_original_code=True):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj_q = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj_kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.use_original_code = use_original_code

    def forward(self, x, seq_len=196, num_frames=8, approx="none", num_landmarks=128):
        out, attn, _ = trajectory_attention(self, x, seq_len, num_frames, approx, None)
        return out, attn


def trajectory_attention(mod, x, P, F, approx="none", size=None):
    """Trajectory attention with exact (approx='none') spatial attention; optional log(size) bias on the
    keys.  Returns (output, temporal attention, per-head keys without the class token)."""
    if approx != "none":
        raise NotImplementedError("only exact trajectory attention (approx='none') is hosted here")
    B, N, C = x.shape
    h = mod.num_heads
    q, k, v = mod.qkv(x).chunk(3, dim=-1)
    q, k, v = (rearrange(t, "b n (h d) -> (b h) n d", h=h) for t in (q, k, v))
    (cls_q, q_), (cls_k, k_), (cls_v, v_) = ((t[:, 0:1], t[:, 1:]) for t in (q, k, v))
    cls_out = rearrange(qkv_attn(cls_q * mod.scale, k, v), "(b h) f d -> b f (h d)", f=1, h=h)
    q_dot_k = rearrange(q_ @ k_.transpose(-2, -1), "b q (f n) -> b q f n", f=F) * mod.scale
    if size is not None:
        q_dot_k = rearrange(q_dot_k, "(b h) q f n -> b h q (f n)", h=h, f=F)
        q_dot_k = q_dot_k + size.log()[:, None, None, :, 0].to(q_dot_k.dtype)
        q_dot_k = rearrange(q_dot_k, "b h q (f n) -> (b h) q f n", h=h, f=F)
    attn = mod.attn_drop(q_dot_k.softmax(dim=-1))
    v_ = rearrange(v_, "b (f n) d -> b f n d", f=F, n=P)
    y = torch.einsum("b q f n, b f n d -> b q f d", attn, v_)
    # temporal attention along the trajectory: the query is the token's own-frame aggregate
    y = rearrange(y, "(b h) s f d -> b s f (h d)", b=B)
    y_diag = rearrange(y, "b (g n) f d -> b g n f d", g=F)
    y_diag = torch.diagonal(y_diag, dim1=-4, dim2=-2)
    y_diag = rearrange(y_diag, "b n d f -> b (f n) d", f=F)
    q2 = rearrange(mod.proj_q(y_diag), "b s (h d) -> b h s d", h=h) * mod.scale
    if mod.use_original_code and type(mod.proj_kv) is nn.Linear and not mod.proj_kv._forward_hooks:
        # v2 is never read on this setting (the sum below runs over y): only the key half of the projection
        pb = mod.proj_kv.bias
        k2 = torch.nn.functional.linear(y, mod.proj_kv.weight[:C], None if pb is None else pb[:C])
        k2, v2 = rearrange(k2, "b s f (h d) -> b h s f d", f=F, h=h), None
    else:
        k2, v2 = mod.proj_kv(y).chunk(2, dim=-1)
        k2, v2 = (rearrange(t, "b s f (h d) -> b h s f d", f=F, h=h) for t in (k2, v2))
    # F = 8 logits per trajectory: a broadcast multiply + reduction streams k2 once; as the batched
    # [1 x d] @ [d x F] products the einsum form lowers to, it is the slowest kernel of the model on MI355X
    tattn = (k2 * q2.unsqueeze(-2)).sum(dim=-1).softmax(dim=-1)
    if mod.use_original_code:
        val = rearrange(y, "b s f (h d) -> b h s f d", f=F, h=h)
    else:
        val = v2
    out = rearrange((val * tattn.unsqueeze(-1)).sum(dim=-2), "b h s d -> b s (h d)")  # same remark
    out = mod.proj_drop(mod.proj(torch.cat((cls_out, out), dim=1)))
    return out, tattn, k_


class Block(nn.Module):
    def __init__(self, dim=768, num_heads=12, mlp_ratio=4.0, qkv_bias=False, use_original_code=True, eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = TrajectoryAttention(dim, num_heads, qkv_bias, use_original_code=use_original_code)
        self.drop_path = nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, seq_len=196, num_frames=8, approx="none", num_landmarks=128):
        x = x + self.attn(self.norm1(x), seq_len=seq_len, num_frames=num_frames, approx=approx)[0]
        return x + self.mlp(self.norm2(x))


class PatchEmbed3D(nn.Module):
    def __init__(self, img_size=224, temporal_resolution=4, in_chans=3, patch_size=16, z_block_size=2, embed_dim=768):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2 * (temporal_resolution // z_block_size)
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=(z_block_size, patch_size, patch_size),
                              stride=(z_block_size, patch_size, patch_size))

    def forward(self, x):
        return tubelet_tokens(self.proj, x)


class Motionformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, patch_size_temp=2, temporal_resolution=8, in_chans=3,
                 num_classes=400, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, use_mlp=True,
                 head_act="tanh", use_original_code=True):
        super().__init__()
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.temporal_resolution = temporal_resolution
        self.spatial_patches = (img_size // patch_size) ** 2
        # input clips carry temporal_resolution * patch_size_temp frames
        self.patch_embed_3d = PatchEmbed3D(img_size, temporal_resolution * patch_size_temp, in_chans, patch_size,
                                           patch_size_temp, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.spatial_patches + 1, embed_dim))
        self.temp_embed = nn.Parameter(torch.zeros(1, temporal_resolution, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, use_original_code)
                                     for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        if use_mlp:
            act = {"tanh": nn.Tanh(), "gelu": nn.GELU()}.get(head_act, nn.ReLU())
            self.pre_logits = nn.Sequential(OrderedDict([("fc", nn.Linear(embed_dim, embed_dim)), ("act", act)]))
        else:
            self.pre_logits = nn.Identity()
        self.head = nn.Linear(embed_dim, num_classes)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        self.apply(self._init)

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
        x = x[0]
        B = x.shape[0]
        x = self.patch_embed_3d(x)  # frame-major tokens '(f n)'
        x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
        npatch = self.spatial_patches
        cls_embed = self.pos_embed[:, 0, :].unsqueeze(1)
        tile_pos = self.pos_embed[:, 1:, :].repeat(1, self.temporal_resolution, 1)
        tile_tmp = self.temp_embed.repeat_interleave(npatch, 1)
        x = x + torch.cat([cls_embed, tile_pos + tile_tmp], dim=1)
        for blk in self.blocks:
            x = blk(x, seq_len=npatch, num_frames=self.temporal_resolution, approx="none", num_landmarks=128)
        return self.pre_logits(self.norm(x)[:, 0])

    def forward(self, x):
        x = self.head(self.forward_features(x))
        if not self.training:
            x = torch.nn.functional.softmax(x, dim=-1)
        return x


def motionformer_base(**kw) -> Motionformer:
    """Motionformer 224 16x4: 8 temporal tokens x 196 patches + cls."""
    return Motionformer(img_size=224, patch_size=16, patch_size_temp=2, temporal_resolution=8, embed_dim=768, depth=12,
                        num_heads=12, **kw)
