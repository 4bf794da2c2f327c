#!/usr/bin/env python3
"""Probe: does the exact-erf GELU pass (HBM bound, few registers) hide beside a library GEMM (persistent Stream-K grid,
one workgroup per CU) on a second stream?  fc1 / fc2 shapes of VideoMAE-B, bf16.  One GEMM grid at a time."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from tome import _abi  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)


def timed(fn, rep=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rep):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / rep * 1e3


for B in (128, 384):
    M = B * 1552
    y = torch.randn(M, 768, device=dev).bfloat16()
    w1 = (torch.randn(3072, 768, device=dev) * 0.03).bfloat16()
    b1 = torch.zeros(3072, device=dev).bfloat16()
    w2 = (torch.randn(768, 3072, device=dev) * 0.03).bfloat16()
    h_other = torch.randn(M, 3072, device=dev).bfloat16()
    keep = {}

    def fc1():
        keep["h"] = F.linear(y, w1, b1)

    def gelu_other():
        _abi.gelu_erf(h_other, inplace=True)

    def fc2_other():
        keep["o"] = F.linear(h_other, w2)

    def both(gemm):
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        gemm()
        with torch.cuda.stream(side):
            gelu_other()
        main.wait_stream(side)

    # chunked: fc1 in n pieces, the GELU of piece i beside the GEMM of piece i + 1
    def chunked(n):
        hs = []
        step = (M // n + 255) // 256 * 256
        for i in range(n):
            lo, hi = i * step, min(M, (i + 1) * step)
            h = F.linear(y[lo:hi], w1, b1)
            hs.append(h)
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                _abi.gelu_erf(h, inplace=True)
        main.wait_stream(side)
        keep["hs"] = hs

    t1, tg, t2 = timed(fc1), timed(gelu_other), timed(fc2_other)
    print(f"batch {B}: fc1 {t1:7.1f} us   gelu {tg:7.1f} us   fc2 {t2:7.1f} us", flush=True)
    print(f"   fc1 || gelu(other buffer) {timed(lambda: both(fc1)):7.1f} us (sum {t1 + tg:7.1f})   "
          f"fc2 || gelu(other buffer) {timed(lambda: both(fc2_other)):7.1f} us (sum {t2 + tg:7.1f})", flush=True)
    for n in (2, 4, 8):
        print(f"   fc1 in {n} pieces, GELU of piece i beside the GEMM of piece i+1: {timed(lambda: chunked(n)):7.1f} us", flush=True)
    del y, h_other, keep
    torch.cuda.empty_cache()
