#!/usr/bin/env python3
"""Where the HOST time of a small-batch patched forward goes (the reference's protocol runs batch 8: the forward is
launch-bound there): cProfile over `iters` forwards, top functions by own time, plus eager vs HIP-graph replay times.
    python tools/host_profile.py timesformer 16 8 [iters]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import motionformer, timesformer, videomae, vivit  # noqa: E402
from hosts.graphed import GraphedForward  # noqa: E402

fam, r, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
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


def run(n, fwd):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fwd(clip)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t_issue / n * 1e3, (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for _ in range(8):
        model(clip)
    issue, total = run(iters, model)
    print(f"{fam} r={r} batch {batch} eager: host issue {issue:.3f} ms / forward, wall {total:.3f} ms / forward "
          f"({batch / total * 1e3:.0f} clips/s)")
    pr = cProfile.Profile()
    pr.enable()
    run(iters, model)
    pr.disable()
    g = GraphedForward(model, clip)
    for _ in range(3):
        g(clip)
    gi, gt = run(iters, g)
    print(f"HIP-graph replay: wall {gt:.3f} ms / forward ({batch / gt * 1e3:.0f} clips/s)")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
