#!/usr/bin/env python3
"""Probe: torch._addmm_activation(bias, x, Wt, use_gelu=True) -- is the tanh-form GELU applied inside the library GEMM
on this build, what does it cost, and how far is it from gelu_fast applied to the rounded dense output?"""
import torch
import torch.nn.functional as F


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


dev = torch.device("cuda", 0)
M, K, N = 64 * 3137, 768, 3072
x = torch.randn(M, K, device=dev).bfloat16()
w = (0.05 * torch.randn(N, K, device=dev)).bfloat16()
b = (0.1 * torch.randn(N, device=dev)).bfloat16()
t_lin = timed(lambda: F.linear(x, w, b))
t_gelu = timed(lambda: F.gelu(F.linear(x, w, b), approximate="tanh"))
t_fused = timed(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True))
y_ref = F.gelu(F.linear(x, w, b).float(), approximate="tanh")
y_f = torch._addmm_activation(b, x, w.t(), use_gelu=True).float()
print(f"linear {t_lin:.1f} us | linear + tanh-gelu pass {t_gelu:.1f} us | _addmm_activation(use_gelu) {t_fused:.1f} us")
print("max |fused - gelu(rounded dense)|:", float((y_f - y_ref).abs().max()), " max |ref|:", float(y_ref.abs().max()))
