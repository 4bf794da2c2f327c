#!/usr/bin/env python3
"""Debug: are repeated replays of one captured forward (and repeated eager forwards) bit-identical?"""
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
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = videomae.videomae_base(16).to(dev).to(torch.bfloat16).eval()
tome.patch.videomae(model, prop_attn=False)
model.r = 16
clip = [torch.rand(batch, 3, 16, 224, 224, device=dev).to(torch.bfloat16)]
_overlap.ENABLED = False
fwd_off = GraphedForward(model, clip)
_overlap.ENABLED = True
fwd_on = GraphedForward(model, clip)
_overlap.ENABLED = False


def distinct(fn):
    outs = []
    for _ in range(n):
        o = fn().clone()
        if not any(torch.equal(o, p) for p in outs):
            outs.append(o)
    return outs


with torch.no_grad():
    e = distinct(lambda: model(clip))
g_off = distinct(lambda: fwd_off(clip))
g_on = distinct(lambda: fwd_on(clip))
print(f"batch {batch}, {n} runs each: distinct outputs -- eager {len(e)}, capture-stream graph {len(g_off)}, side-stream graph {len(g_on)}")
print(f"   graph(off)[0] == eager[0]: {torch.equal(g_off[0], e[0])}; graph(on)[0] == eager[0]: {torch.equal(g_on[0], e[0])}")
for name, outs in (("eager", e), ("graph off", g_off), ("graph on", g_on)):
    if len(outs) > 1:
        print(f"   {name}: max diff between variants {max(float((o.float() - outs[0].float()).abs().max()) for o in outs[1:]):.3g}")
