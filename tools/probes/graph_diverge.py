#!/usr/bin/env python3
"""Debug: which op of a captured VideoMAE forward first differs from the eager forward (capture order on,off)?  Hooks
keep references to the capture's own intermediate tensors (kept alive, so the graph's pool cannot reuse them); after a
replay they hold that replay's values."""
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
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = videomae.videomae_base(16).to(dev).to(torch.bfloat16).eval()
tome.patch.videomae(model, prop_attn=False)
model.r = 16
clip = [torch.rand(batch, 3, 16, 224, 224, device=dev).to(torch.bfloat16)]
store = {"mode": None, "eager": [], "graph": []}


def hook(name):
    def h(m, a, o):
        t = o[0] if isinstance(o, tuple) else o
        if store["mode"] == "eager":
            store["eager"].append((name, t.clone()))
        elif store["mode"] == "graph":
            store["graph"].append((name, t))  # a reference: the capture's own tensor
    return h


for i, blk in enumerate(model.model.blocks):
    blk.register_forward_hook(hook(f"block{i}"))
    blk.attn.register_forward_hook(hook(f"block{i}.attn"))
    blk.attn.qkv.register_forward_hook(hook(f"block{i}.attn.qkv")) if hasattr(blk.attn.qkv, "register_forward_hook") else None
    blk.attn.proj.register_forward_hook(hook(f"block{i}.attn.proj"))
    blk.mlp.fc1.register_forward_hook(hook(f"block{i}.mlp.fc1"))
model.model.patch_embed.register_forward_hook(hook("patch_embed"))
if len(sys.argv) > 2 and sys.argv[2] == "on_first":
    _overlap.ENABLED = True
    first = GraphedForward(model, clip)  # an ON capture first
_overlap.ENABLED = False
fwd = GraphedForward.__new__(GraphedForward)
# capture by hand so that the hooks see only the capture pass
fwd.model = model
fwd.static_in = [t.clone() for t in clip]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.no_grad(), torch.cuda.stream(side):
    for _ in range(3):
        model(fwd.static_in)
torch.cuda.current_stream().wait_stream(side)
fwd.graph = torch.cuda.CUDAGraph()
store["mode"] = "graph"
with torch.no_grad(), torch.cuda.graph(fwd.graph):
    fwd.static_out = model(fwd.static_in)
store["mode"] = "eager"
with torch.no_grad():
    eager = model(clip).clone()
store["mode"] = None
out = fwd(clip).clone()
torch.cuda.synchronize()
print(f"replay == eager: {torch.equal(out, eager)} (max diff {float((out.float() - eager.float()).abs().max()):.3g}); "
      f"{len(store['graph'])} tensors kept from the capture, {len(store['eager'])} from the eager run")
for (n1, g), (n2, e) in zip(store["graph"], store["eager"]):
    assert n1 == n2
    if g.shape != e.shape or not torch.equal(g, e):
        d = float((g.float() - e.float()).abs().max()) if g.shape == e.shape else None
        print(f"   first difference: {n1} {tuple(g.shape)} max diff {d}; differing elements {int((g != e).sum()) if g.shape == e.shape else None}")
        break
else:
    print("   no kept tensor differs")
