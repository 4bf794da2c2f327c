#!/usr/bin/env python3
"""Debug: does a patched forward read memory it never wrote?  The caching allocator's pool is pre-filled with a byte
pattern (one big tensor filled and released: every later torch.empty is carved out of it); the logits must not depend
on the pattern.    python3 tools/probes/poison_forward.py timesformer 16 64"""
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
_overlap.ENABLED = len(sys.argv) > 4 and sys.argv[4] == "on"
plans = []
real_match = _abi.match_keys


def spy(*a, **k):
    p = real_match(*a, **k)
    plans.append(p)
    return p


_abi.match_keys = spy
with torch.no_grad():
    model(clip)
torch.cuda.synchronize()
peak = torch.cuda.max_memory_allocated()
outs = {}
for fill in (0, 255, 0x7F, "rand", 0):
    torch.cuda.empty_cache()
    junk = torch.empty(int(peak * 1.3), dtype=torch.uint8, device=dev)
    if fill == "rand":
        junk.random_(0, 256)
    else:
        junk.fill_(fill)
    torch.cuda.synchronize()
    del junk
    plans.clear()
    with torch.no_grad():
        out = model(clip).float().clone()
    torch.cuda.synchronize()
    got = [(p.src_idx.clone(), p.dst_idx.clone(), p.unm_idx.clone()) for p in plans]
    if not outs:
        outs["base"] = (out, got)
        print(f"{fam} r={r} batch {batch} (side stream {'on' if _overlap.ENABLED else 'off'}), pool pre-filled with {fill}: base", flush=True)
        continue
    b_out, b_got = outs["base"]
    first_bad = next((i for i, (a, b) in enumerate(zip(b_got, got)) if not all(torch.equal(x, y) for x, y in zip(a, b))), None)
    print(f"   pool pre-filled with {fill}: logits equal {torch.equal(out, b_out)} (max diff {float((out - b_out).abs().max()):.3g}, "
          f"nan {int(out.isnan().sum())}); first layer whose plan differs: {first_bad}", flush=True)
