#!/usr/bin/env python3
"""Probe: the library GEMM with the tanh-GELU epilogue (torch._addmm_activation -> hipBLASLt GELU_BIAS) against
F.linear + F.gelu(approximate='tanh') on ViViT's fc1 shape (bf16): time and difference."""
import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
torch.manual_seed(0)
for M in (64 * 3137, 8 * 3137):
    x = torch.randn(M, 768, device=dev).bfloat16()
    w = (torch.randn(3072, 768, device=dev) * 0.03).bfloat16()
    b = (torch.randn(3072, device=dev) * 0.1).bfloat16()

    def two():
        return F.gelu(F.linear(x, w, b), approximate="tanh")

    def one():
        return torch._addmm_activation(b, x, w.t(), use_gelu=True)

    def timed(fn, rep=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(rep):
            fn()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) / rep * 1e3

    y2, y1 = two(), one()
    ref = F.gelu(F.linear(x.float(), w.float(), b.float()), approximate="tanh")
    print(f"M={M}: linear+gelu {timed(two):8.1f} us   _addmm_activation {timed(one):8.1f} us   linear alone "
          f"{timed(lambda: F.linear(x, w, b)):8.1f} us", flush=True)
    print(f"   max |one - two| {float((y1.float() - y2.float()).abs().max()):.4g}   vs fp32: one {float((y1.float() - ref).abs().max()):.4g} "
          f"two {float((y2.float() - ref).abs().max()):.4g}   differing elements {float((y1 != y2).float().mean()):.4f}", flush=True)
    del x, y1, y2, ref
