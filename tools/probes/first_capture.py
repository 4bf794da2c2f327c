#!/usr/bin/env python3
"""Debug: the FIRST capture of a process (no eager forward before it) against the eager forward, VideoMAE batch 8."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import videomae  # noqa: E402
from hosts.graphed import GraphedForward  # noqa: E402
from tome import _overlap  # noqa: E402

r = int(sys.argv[1]) if len(sys.argv) > 1 else 16
patched = not (len(sys.argv) > 2 and sys.argv[2] == "unpatched")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = videomae.videomae_base(16).to(dev).to(torch.bfloat16).eval()
if patched:
    tome.patch.videomae(model, prop_attn=False)
    model.r = r
clip = [torch.rand(8, 3, 16, 224, 224, device=dev).to(torch.bfloat16)]
_overlap.ENABLED = False
fwd = GraphedForward(model, clip)
with torch.no_grad():
    eager = model(clip).clone()
out = fwd(clip).clone()
fwd2 = GraphedForward(model, clip)
out2 = fwd2(clip).clone()
sw = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("TOME_"))
print(f"r={r} {'patched' if patched else 'UNPATCHED host model'} [{sw}]: first capture == eager {torch.equal(out, eager)} "
      f"(max diff {float((out.float() - eager.float()).abs().max()):.3g}); second capture == eager {torch.equal(out2, eager)}", flush=True)
