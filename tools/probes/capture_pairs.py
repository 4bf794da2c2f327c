#!/usr/bin/env python3
"""Debug: several captures of one forward in a process with no eager forward before them; every replay against the
eager forward run afterwards.  argv: r (0 = no merging) | 'unpatched', then a capture order like off,on,off."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import videomae  # noqa: E402
from hosts.graphed import GraphedForward  # noqa: E402
from tome import _overlap  # noqa: E402

what = sys.argv[1]
order = sys.argv[2].split(",")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = videomae.videomae_base(16).to(dev).to(torch.bfloat16).eval()
if what != "unpatched":
    tome.patch.videomae(model, prop_attn=False)
    model.r = int(what)
clip = [torch.rand(8, 3, 16, 224, 224, device=dev).to(torch.bfloat16)]
fwds = []
for tag in order:
    _overlap.ENABLED = tag == "on"
    fwds.append(GraphedForward(model, clip))
_overlap.ENABLED = False
with torch.no_grad():
    eager = model(clip).clone()
outs = [f(clip).clone() for f in fwds]
print(f"{what} captures {order}: replay == eager {[bool(torch.equal(o, eager)) for o in outs]}; "
      f"max diffs {[round(float((o.float() - eager.float()).abs().max()), 5) for o in outs]}", flush=True)
