#!/usr/bin/env python3
"""Debug: replay of a captured forward before and after <something else> happens in the process."""
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
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = videomae.videomae_base(16).to(dev).to(torch.bfloat16).eval()
if os.environ.get("UNPATCHED") != "1":
    tome.patch.videomae(model, prop_attn=False)
    model.r = 16
clip = [torch.rand(8, 3, 16, 224, 224, device=dev).to(torch.bfloat16)]
_overlap.ENABLED = False
fwd = GraphedForward(model, clip)
a = fwd(clip).clone()
a2 = fwd(clip).clone()
if what == "capture_on":
    _overlap.ENABLED = True
    other = GraphedForward(model, clip)
elif what == "capture_off":
    other = GraphedForward(model, clip)
elif what == "eager_on_forced":
    _overlap.ENABLED = True
    _overlap.MIN_WORK = 0
    with torch.no_grad():
        model(clip)
elif what == "eager_off":
    with torch.no_grad():
        model(clip)
elif what == "side_objects":
    _overlap._state(dev)
elif what == "capture_on_other_model":
    torch.manual_seed(1)
    m2 = videomae.videomae_base(16).to(dev).to(torch.bfloat16).eval()
    tome.patch.videomae(m2, prop_attn=False)
    m2.r = 16
    _overlap.ENABLED = True
    other = GraphedForward(m2, clip)
torch.cuda.synchronize()
_overlap.ENABLED = False
b = fwd(clip).clone()
with torch.no_grad():
    e = model(clip).clone()
print(f"{what}{' (UNPATCHED host model: framework ops only)' if os.environ.get('UNPATCHED') == '1' else ''}: replay == replay again {torch.equal(a, a2)}; replay before == replay after {torch.equal(a, b)} "
      f"(max diff {float((a.float() - b.float()).abs().max()):.3g}); before == eager {torch.equal(a, e)}; after == eager {torch.equal(b, e)}", flush=True)
