#!/usr/bin/env python3
"""Debug: does tome_match_keys (filter path) depend on what its workspace held before?  Keys of TimeSformer's layers at
batch 64 (n = 512 groups), the workspace carved out of memory pre-filled with several byte patterns; the plan must not
move, and must equal the plan of the fp32 all-pairs pass (TOME_SCORES_FILTER=0)."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import timesformer  # noqa: E402
from tome import _abi, _overlap  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = timesformer.timesformer_base(8).to(dev).to(torch.bfloat16).eval()
tome.patch.timesformer(model)
model.r = 16
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
clip = [torch.rand(batch, 3, 8, 224, 224, device=dev).to(torch.bfloat16)]
_overlap.ENABLED = False
grabbed = []
real = _abi.match_keys


def spy(keys, r, *a, **k):
    grabbed.append((keys.clone(), r))
    return real(keys, r, *a, **k)


_abi.match_keys = spy
with torch.no_grad():
    model(clip)
torch.cuda.synchronize()
_abi.match_keys = real


def same(p, q):
    return torch.equal(p.src_idx, q.src_idx) and torch.equal(p.dst_idx, q.dst_idx) and torch.equal(p.unm_idx, q.unm_idx)


def with_fill(keys, r, fill):
    nbytes = 1 << 30
    junk = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if fill == "rand":
        junk.random_(0, 256)
    else:
        junk.fill_(fill)
    torch.cuda.synchronize()
    del junk  # back to the pool: the next workspace is carved out of it
    return real(keys, r, want_node_max=True)


for layer, (keys, r) in enumerate(grabbed[:6]):
    os.environ["TOME_SCORES_FILTER"] = "0"
    ref = real(keys, r, want_node_max=True)
    os.environ.pop("TOME_SCORES_FILTER")
    res = {f: with_fill(keys, r, f) for f in (0, 255, 0x7F, "rand")}
    torch.cuda.synchronize()
    line = f"layer {layer}: keys {tuple(keys.shape)} r={r}:"
    for f, p in res.items():
        bad = (p.node_max != ref.node_max) & ~(p.node_max.isnan() & ref.node_max.isnan())
        line += f"  fill {f}: plan == fp32 pass {same(p, ref)} (node_max differs in {int(bad.sum())} rows)"
    print(line, flush=True)
