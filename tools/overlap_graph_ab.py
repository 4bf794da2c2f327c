#!/usr/bin/env python3
"""A/B of tome/_overlap.py under HIP-graph replay (reference protocol, batch 8): the captured forward with the layer's
matching on the side stream (on) and on the capture stream (off).
    python3 tools/overlap_graph_ab.py timesformer 16 8"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import motionformer, timesformer, videomae, vivit  # noqa: E402
from hosts.graphed import GraphedForward  # noqa: E402
from tome import _overlap  # noqa: E402

fam, r, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
build, patch, frames, kw = {
    "videomae": (lambda: videomae.videomae_base(16), tome.patch.videomae, 16, {"prop_attn": False}),
    "timesformer": (lambda: timesformer.timesformer_base(8), tome.patch.timesformer, 8, {}),
    "motionformer": (lambda: motionformer.motionformer_base(), tome.patch.motionformer, 16, {}),
    "vivit": (lambda: vivit.vivit_base(32), tome.patch.vivit, 32, {}),
}[fam]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = build().to(dev).to(torch.bfloat16).eval()
patch(model, **kw)
model.r = r
clip = [torch.rand(batch, 3, frames, 224, 224, device=dev).to(torch.bfloat16)]
fwd = {}
for on in (False, True):
    _overlap.ENABLED = on
    fwd[on] = GraphedForward(model, clip)
with torch.no_grad():
    _overlap.ENABLED = False
    eager = model(clip).clone()
out_on, out_off = fwd[True](clip).clone(), fwd[False](clip).clone()
print(f"replay == eager forward: capture-stream graph {torch.equal(out_off, eager)}, side-stream graph {torch.equal(out_on, eager)} "
      f"(max diff {float((out_on.float() - eager.float()).abs().max()):.3g}); again: {torch.equal(fwd[True](clip), eager)}", flush=True)
best = {True: 0.0, False: 0.0}
for rnd in range(4):
    for on in (False, True):
        fwd[on](clip)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            fwd[on](clip)
        b.record()
        torch.cuda.synchronize()
        best[on] = max(best[on], batch * 40 / (a.elapsed_time(b) * 1e-3))
print(f"{fam} r={r} batch {batch} HIP-graph replay: matching on the capture stream {best[False]:8.1f} clips/s   on the side "
      f"stream {best[True]:8.1f}   x{best[True] / best[False]:.3f}", flush=True)
