#!/usr/bin/env python3
"""Probe: the block's second residual inside the fc2 GEMM (`x.addmm_(h, W2ᵀ)`, beta = 1 on the residual buffer) against
`F.linear(h, W2, b2)` + a separate add -- what does the library GEMM cost with and without the C operand?"""
import sys
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


def main():
    dev = torch.device("cuda", 0)
    for M in (128 * 1568, 128 * 1392, 64 * 1568):
        for K, N in ((3072, 768), (768, 768)):
            h = torch.randn(M, K, device=dev).bfloat16()
            w = (0.02 * torch.randn(N, K, device=dev)).bfloat16()
            b = torch.randn(N, device=dev).bfloat16()
            x = torch.randn(M, N, device=dev).bfloat16()
            out = torch.empty_like(x)
            t_lin = timed(lambda: F.linear(h, w, b))
            t_lin_out = timed(lambda: torch.addmm(b, h, w.t(), out=out))
            t_inpl = timed(lambda: x.addmm_(h, w.t()))
            t_add = timed(lambda: torch.add(x, out))
            fl = 2.0 * M * K * N
            print(f"M={M} K={K} N={N}: linear+bias {t_lin:.1f} us ({fl / t_lin / 1e6:.0f} TF)  addmm(out=) {t_lin_out:.1f}  "
                  f"x.addmm_ (beta=1) {t_inpl:.1f} us ({fl / t_inpl / 1e6:.0f} TF)  separate add {t_add:.1f} us", flush=True)


if __name__ == "__main__":
    sys.exit(main())
