#!/usr/bin/env python3
"""Merge-path micro-benchmark at any (batch, tokens, r, class_token): per-kernel HIP-event timings over the
12-layer chain, as bench.py measures them for the headline workload.
    python tools/bench_path.py --batch 32 --tokens 1568 --r 150
    python tools/bench_path.py --batch 8 --tokens 3137 --r 300 --cls"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--tokens", type=int, default=1568)
ap.add_argument("--r", type=int, default=16)
ap.add_argument("--layers", type=int, default=12)
a = ap.parse_args()
bench.LAYERS = a.layers
with torch.no_grad():
    st = bench.measure_kernels(a.batch, a.tokens, a.r, torch.device("cuda", 0), reps=5)
tot = sum(v["ms"] for v in st.values())
print(f"batch {a.batch} tokens {a.tokens} r {a.r}: merge path {tot:.3f} ms per forward")
for k, v in st.items():
    extra = f"{v['bytes'] / (v['ms'] / 1e3) / 1e9:8.0f} GB/s" if v["ms"] else ""
    if v["flops"]:
        extra += f" {v['flops'] / (v['ms'] / 1e3) / 1e12:6.1f} TF/s"
    print(f"  {k:18s} {v['ms'] * 1e3 / max(1, v['launches']):8.1f} us/launch x{v['launches']:2d}  {extra}")
