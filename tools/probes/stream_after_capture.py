#!/usr/bin/env python3
"""Platform probe (nothing of this package in the graph when UNPATCHED=1): a captured forward replayed before and after
a NEW HIP stream is created in the process.    python3 tools/probes/stream_after_capture.py stream|events|nothing"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import videomae  # noqa: E402

what = sys.argv[1]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = videomae.videomae_base(16).to(dev).to(torch.bfloat16).eval()
if os.environ.get("UNPATCHED") != "1":
    tome.patch.videomae(model, prop_attn=False)
    model.r = 16
    if os.environ.get("PREPARE") == "1":
        from tome import _overlap
        _overlap._state(dev)
clip = [torch.rand(8, 3, 16, 224, 224, device=dev).to(torch.bfloat16)]
static_in = [clip[0].clone()]
warm = torch.cuda.Stream()
warm.wait_stream(torch.cuda.current_stream())
with torch.no_grad(), torch.cuda.stream(warm):
    for _ in range(3):
        model(static_in)
torch.cuda.current_stream().wait_stream(warm)
g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g):
    static_out = model(static_in)
g.replay()
a = static_out.clone()
torch.cuda.synchronize()
keep = []
if what == "stream":
    keep.append(torch.cuda.Stream(device=dev))
elif what == "stream_used":
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        keep.append(torch.zeros(4, device=dev) + 1)
    keep.append(s)
elif what == "events":
    for _ in range(2):
        e = torch.cuda.Event()
        e.record()
        keep.append(e)
torch.cuda.synchronize()
outs = []
for _ in range(3):
    g.replay()
    outs.append(static_out.clone())
torch.cuda.synchronize()
with torch.no_grad():
    e = model(clip).clone()
tag = "UNPATCHED host model" if os.environ.get("UNPATCHED") == "1" else "patched" + (" (side stream made before the capture)" if os.environ.get("PREPARE") == "1" else "")
print(f"{tag}, then '{what}': replay before == eager {torch.equal(a, e)}; replays after == before "
      f"{[bool(torch.equal(o, a)) for o in outs]}; nan in replays after {[int(o.isnan().sum()) for o in outs]}", flush=True)
