#!/usr/bin/env python3
"""Host cost of the fork / join of tome/_overlap.py and the reference protocol (batch 8, one synchronised forward per
iteration: slowfast/utils/model_benchmark.py:20-58) with the matching on the caller's stream / on the side stream.
    python3 tools/probes/overlap_host_cost.py [family r batch]"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import videomae, vivit  # noqa: E402
from tome import _overlap  # noqa: E402

dev = torch.device("cuda", 0)
fam = sys.argv[1] if len(sys.argv) > 1 else "videomae"
r = int(sys.argv[2]) if len(sys.argv) > 2 else 16
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8


def host_us(fn, n=2000):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t) / n * 1e6
    torch.cuda.synchronize()
    return dt


st = _overlap._state(dev)
main = torch.cuda.current_stream(dev)
print(f"host cost per call: event.record {host_us(lambda: st[2].record(main)):.1f} us, stream.wait_event "
      f"{host_us(lambda: st[0].wait_event(st[2])):.1f}, current_stream {host_us(lambda: torch.cuda.current_stream(dev)):.1f}, "
      f"new Event + record {host_us(lambda: torch.cuda.Event().record(main)):.1f}, stream context "
      f"{host_us(lambda: torch.cuda.stream(st[0]).__enter__() or torch.cuda.stream(main).__enter__()):.1f} (enter x2)", flush=True)

build, patch, frames, kw = {
    "videomae": (lambda: videomae.videomae_base(16), tome.patch.videomae, 16, {"prop_attn": False}),
    "vivit": (lambda: vivit.vivit_base(32), tome.patch.vivit, 32, {}),
}[fam]
torch.manual_seed(0)
model = build().to(dev).to(torch.bfloat16).eval()
patch(model, **kw)
model.r = r
shape = (batch, 3, frames, 224, 224)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    for rnd in range(3):
        for on in (False, True):
            _overlap.ENABLED = on
            dev_ms, host_ms = [], []
            for it in range(45):
                clip = [torch.rand(shape, device=dev).to(torch.bfloat16)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                a.record()
                model(clip)
                b.record()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                if it >= 5:
                    dev_ms.append(a.elapsed_time(b))
                    host_ms.append((t1 - t0) * 1e3)
            m = sum(dev_ms) / len(dev_ms)
            print(f"{fam} r={r} batch {batch} side stream {'on ' if on else 'off'}: {m:.3f} ms per forward = "
                  f"{batch / m * 1e3:7.1f} clips/s   host issue {sum(host_ms) / len(host_ms):.3f} ms", flush=True)
