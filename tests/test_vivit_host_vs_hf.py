"""Structural cross-check of the ViViT host model (hosts/vivit.py) against the ViViT implementation of the
installed HF `transformers` (third party, 5.x: `VivitForVideoClassification`): same weights -> same logits.

This does NOT pin the reference's ToMe patch for ViViT (tome/patch/vivit.py needs `VivitSelfAttention`, which
this transformers release no longer has -- SURVEY.md 8c: block-level parity of the ViViT *patch* stays unpinned);
it pins the architecture the patch sits in: tubelet embedding, class token, learned positions, pre-LN layers
with gelu_fast, final layernorm, logits from token 0.  CPU only."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]

try:
    from transformers import VivitConfig as HFConfig
    from transformers import VivitForVideoClassification as HFViViT
except Exception:  # pragma: no cover - transformers without ViViT
    HFViViT = None

from hosts.vivit import ViViT, VivitConfig  # noqa: E402


def _rename(k: str) -> str:
    k = k.replace("vivit.layers.", "vivit.encoder.layer.")
    for a, b in ((".attention.q_proj.", ".attention.attention.query."), (".attention.k_proj.", ".attention.attention.key."),
                 (".attention.v_proj.", ".attention.attention.value."), (".attention.o_proj.", ".attention.output.dense."),
                 (".mlp.fc1.", ".intermediate.dense."), (".mlp.fc2.", ".output.dense.")):
        k = k.replace(a, b)
    return k


@pytest.mark.skipif(HFViViT is None, reason="installed transformers has no ViViT")
@pytest.mark.parametrize("frames,size,tubelet,hidden,layers,heads", [(8, 32, (2, 8, 8), 32, 2, 2), (4, 48, (2, 16, 16), 48, 3, 4)])
def test_host_vivit_equals_hf_vivit(frames, size, tubelet, hidden, layers, heads):
    torch.manual_seed(0)
    hf = HFViViT(HFConfig(image_size=size, num_frames=frames, tubelet_size=list(tubelet), hidden_size=hidden,
                          num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=2 * hidden,
                          num_labels=5, hidden_act="gelu_fast", layer_norm_eps=1e-6, qkv_bias=True,
                          attn_implementation="eager")).eval()
    with torch.no_grad():  # HF initialises biases / cls token to zero: randomise everything so each term counts
        for p in hf.parameters():
            p.copy_(torch.randn_like(p) * 0.1)
    host = ViViT(VivitConfig(image_size=size, num_frames=frames, tubelet_size=tubelet, hidden_size=hidden,
                             num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=2 * hidden,
                             layer_norm_eps=1e-6, qkv_bias=True), num_classes=5).eval()
    sd = {_rename(k): v for k, v in hf.state_dict().items()}
    missing, unexpected = host.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    clip = torch.rand(2, 3, frames, size, size)
    with torch.no_grad():
        want = hf(pixel_values=clip.permute(0, 2, 1, 3, 4)).logits
        got = host([clip])
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
