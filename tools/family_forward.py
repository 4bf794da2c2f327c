#!/usr/bin/env python3
"""A few patched forwards of one model family, for a kernel trace:
    cd /tmp && rocprofv3 --kernel-trace -d OUT -- python3 tools/family_forward.py timesformer 32 64 [iters]
    python tools/rocpd_kernels.py OUT --dispatches k_merge_rows_fast      # every launch, in order, with its duration
bf16, random init, synthetic clips resident on the device (the `also` workloads of bench.py)."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import motionformer, timesformer, videomae, vivit  # noqa: E402

fam, r, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 6
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
with torch.no_grad():
    for _ in range(iters):
        model(clip)
        torch.cuda.synchronize()
print(f"{fam} r={r} batch {batch}: {iters} forwards done")
