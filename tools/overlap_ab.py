#!/usr/bin/env python3
"""A/B of tome/_overlap.py inside one process: patched bf16 forwards of one family with the layer's matching on the side
stream (on) and on the caller's stream (off), alternating, a few rounds each; clips/s per setting and the ratio.
    python3 tools/overlap_ab.py videomae 16 8 128 384      # family r batch [batch ...]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import motionformer, timesformer, videomae, vivit  # noqa: E402
from tome import _overlap  # noqa: E402

fam, r = sys.argv[1], int(sys.argv[2])
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
for batch in [int(a) for a in sys.argv[3:]]:
    clip = [torch.rand(batch, 3, frames, 224, 224, device=dev).to(torch.bfloat16)]
    iters = max(4, min(40, 2048 // batch))
    best = {True: 0.0, False: 0.0}
    with torch.no_grad():
        for _ in range(3):
            model(clip)
        for rnd in range(4):
            for on in (False, True):
                _overlap.ENABLED = on
                model(clip)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(iters):
                    model(clip)
                b.record()
                torch.cuda.synchronize()
                best[on] = max(best[on], batch * iters / (a.elapsed_time(b) * 1e-3))
    print(f"{fam} r={r} batch {batch:4d}: matching on the caller's stream {best[False]:8.1f} clips/s   on the side stream "
          f"{best[True]:8.1f}   x{best[True] / best[False]:.3f}", flush=True)
    del clip
    torch.cuda.empty_cache()
