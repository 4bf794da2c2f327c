import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch
from tome import _abi
DEV = "cuda:0"
for (B, H, P, F) in [(1, 12, 64, 8), (2, 3, 196, 8), (1, 2, 36, 4), (2, 1, 100, 3), (4, 12, 64, 8), (4, 12, 32, 8), (4, 12, 96, 8)]:
    for seed in (P * 31 + F, 1, 2):
        g = torch.Generator(device=DEV).manual_seed(seed)
        N = 1 + P * F
        qkv = torch.randn(B, N, 3, H, 64, device=DEV, generator=g).to(torch.bfloat16)
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        qs, ks, vs = q[:, :, 1:], k[:, :, 1:], v[:, :, 1:]
        logits = (qs.float() @ ks.float().transpose(-1, -2)) * 0.125
        w = logits.reshape(B, H, N - 1, F, P).softmax(-1)
        want = torch.einsum("b h q f n, b h f n d -> b q f h d", w, vs.float().reshape(B, H, F, P, 64)).reshape(B, N - 1, F, H * 64)
        errs = {}
        for env in ("0", "1"):
            os.environ["TOME_ATTN_RESIDENT"] = env
            y = _abi.prop_attention_segments(qs, ks, vs, F, 0.125)
            d = (y.float() - want).abs()
            errs[env] = (float(d.max()), float(d.mean()))
        print((B, H, P, F), seed, "streaming max/mean %.4f %.5f | resident %.4f %.5f" % (errs["0"] + errs["1"]))
