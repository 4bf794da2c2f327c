#!/usr/bin/env python3
"""Debug: is a forward with the matching on the side stream bit-identical to one without, at sizes where the matching
takes the filter path (>= 1024 A tiles)?  Compares logits and every layer's plan; optionally synchronises after the fork."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import motionformer, timesformer, videomae, vivit  # noqa: E402
from tome import _abi, _overlap  # noqa: E402

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
_overlap.MIN_WORK = 0
real_ready = _overlap.keys_ready
from tome.patch import _common  # noqa: E402
_common.keys_ready = lambda keys, info, capture_only=False: real_ready(keys, info, False)  # fork in eager mode everywhere
plans = []
real_match = _abi.match_keys


MAIN = torch.cuda.current_stream(dev).cuda_stream


def spy(*a, **k):
    p = real_match(*a, **k)
    cur = torch.cuda.current_stream(dev).cuda_stream
    plans.append((p, None if cur == MAIN else cur))
    return p


_abi.match_keys = spy
real_beside = _overlap.match_beside
sync_after = [False]


def beside(metric, ready, info):
    real_beside(metric, ready, info)
    if sync_after[0]:
        torch.cuda.synchronize()


_common.match_beside = beside


def run(on, sync=False):
    _overlap.ENABLED = on
    sync_after[0] = sync
    plans.clear()
    with torch.no_grad():
        out = model(clip).float().clone()
    torch.cuda.synchronize()
    got = [(p.src_idx.clone(), p.dst_idx.clone(), p.unm_idx.clone(), s) for p, s in plans]
    return out, got


base, p0 = run(False)
base2, p02 = run(False)
print(f"{fam} r={r} batch {batch}: off vs off again: logits equal {torch.equal(base, base2)}", flush=True)
for label, on, sync in (("side stream", True, False), ("side stream + device sync after the fork", True, True), ("side stream again", True, False)):
    out, p1 = run(on, sync)
    side = sum(1 for x in p1 if x[3] is not None)
    first_bad = next((i for i, (a, b) in enumerate(zip(p0, p1)) if not (torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]))), None)
    print(f"   {label}: {side}/{len(p1)} matchings on the side stream; logits equal {torch.equal(out, base)} "
          f"(max diff {float((out - base).abs().max()):.3g}); first layer whose plan differs: {first_bad}", flush=True)
