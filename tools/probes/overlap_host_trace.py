#!/usr/bin/env python3
"""Host issue time per forward over many iterations of the reference protocol, side stream on / off in long runs."""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import videomae  # noqa: E402
from tome import _overlap  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = videomae.videomae_base(16).to(dev).to(torch.bfloat16).eval()
tome.patch.videomae(model, prop_attn=False)
model.r = 16
shape = (8, 3, 16, 224, 224)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
fresh = len(sys.argv) > 1 and sys.argv[1] == "fresh"
with torch.no_grad():
    for on, n in ((False, 60), (True, 200), (False, 60), (True, 100)):
        _overlap.ENABLED = on
        dev_ms, host_ms = [], []
        for it in range(n):
            clip = [torch.rand(shape, device=dev).to(torch.bfloat16)]
            torch.cuda.synchronize()
            if fresh and on:
                _overlap._side.clear()
            t0 = time.perf_counter()
            a.record()
            model(clip)
            b.record()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            dev_ms.append(a.elapsed_time(b))
            host_ms.append((t1 - t0) * 1e3)
            if (it + 1) % 20 == 0:
                print(f"side stream {'on ' if on else 'off'} it {it - 19:3d}-{it:3d}: forward {sum(dev_ms[-20:]) / 20:.3f} ms   "
                      f"host issue {sum(host_ms[-20:]) / 20:.3f} ms", flush=True)
