#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (parity-test cases, not the headline bench line):
TimeSformer divST 8x224 r in {0,8,16,32}, ViViT-B 32x224 (3137 tokens) r in {0,64}, Motionformer 224 16x4
r in {0,16}, VideoMAE-B 8x224 / 16x224.  bf16, random init, synthetic clips, one GPU.
    python tools/bench_models.py [--batch 8] [--iters 10]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import tome  # noqa: E402
from hosts import motionformer, timesformer, videomae, vivit  # noqa: E402


def run(name, build, patch, frames, r_values, batch, iters, dev):
    out = []
    for r in r_values:
        torch.manual_seed(0)
        model = build().to(dev).to(torch.bfloat16).eval()
        patch(model)
        model.r = r
        clip = [torch.rand(batch, 3, frames, 224, 224, device=dev).to(torch.bfloat16)]
        with torch.no_grad():
            for _ in range(8):  # every layer has its own token count: let the GEMM / attention heuristics settle
                model(clip)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                model(clip)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        rec = {"model": name, "r": r, "batch": batch, "clips_per_s": round(batch / dt, 1), "ms": round(dt * 1e3, 2)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del model
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    res = []
    cfgs = [
        ("VideoMAE-B 16x224", lambda: videomae.videomae_base(16), tome.patch.videomae, 16, (0, 8, 16, 150)),
        ("VideoMAE-B 8x224", lambda: videomae.videomae_base(8), tome.patch.videomae, 8, (0, 16)),
        ("TimeSformer divST 8x224", lambda: timesformer.timesformer_base(8), tome.patch.timesformer, 8, (0, 8, 16, 32)),
        ("Motionformer 224 16x4", lambda: motionformer.motionformer_base(), tome.patch.motionformer, 16, (0, 16)),
        ("ViViT-B 32x224", lambda: vivit.vivit_base(32), tome.patch.vivit, 32, (0, 64, 300)),
    ]
    for name, build, patch, frames, rs in cfgs:
        if a.only and a.only.lower() not in name.lower():
            continue
        res += run(name, build, patch, frames, rs, a.batch, a.iters, dev)
    with open(os.path.join(ROOT, "gpurun_out", "bench_models.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
