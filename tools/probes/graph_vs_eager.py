#!/usr/bin/env python3
"""Debug: HIP-graph replays of the patched VideoMAE forward against the eager forward, captured several times."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import videomae  # noqa: E402
from hosts.graphed import GraphedForward  # noqa: E402
from tome import _overlap  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
order = sys.argv[2] if len(sys.argv) > 2 else "off,off,on,off"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = videomae.videomae_base(16).to(dev).to(torch.bfloat16).eval()
tome.patch.videomae(model, prop_attn=False)
model.r = 16
clip = [torch.rand(batch, 3, 16, 224, 224, device=dev).to(torch.bfloat16)]
with torch.no_grad():
    _overlap.ENABLED = False
    eager0 = model(clip).clone()
fwds = []
for tag in order.split(","):
    _overlap.ENABLED = tag == "on"
    fwds.append((tag, GraphedForward(model, clip)))
with torch.no_grad():
    _overlap.ENABLED = False
    eager1 = model(clip).clone()
print(f"batch {batch}: eager before the captures == eager after: {torch.equal(eager0, eager1)}")
for i, (tag, f) in enumerate(fwds):
    out = f(clip).clone()
    print(f"   capture {i} ({tag}): replay == eager {torch.equal(out, eager1)}  max diff {float((out.float() - eager1.float()).abs().max()):.3g}", flush=True)
