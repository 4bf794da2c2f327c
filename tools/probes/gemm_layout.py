#!/usr/bin/env python3
"""Probe: does the library GEMM of the block's four projections run faster from a pre-transposed weight
(`x @ Wt`, Wt = W.t().contiguous(), "NN") than from nn.Linear's [N, K] weight ("TN")?  bf16, M = batch x tokens."""
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
    for M in (128 * 1568, 128 * 1392):
        for K, N in ((768, 2304), (768, 768), (768, 3072), (3072, 768)):
            x = torch.randn(M, K, device=dev).bfloat16()
            w = (0.02 * torch.randn(N, K, device=dev)).bfloat16()
            wt = w.t().contiguous()
            b = torch.randn(N, device=dev).bfloat16()
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            fl = 2.0 * M * K * N
            t_tn = timed(lambda: F.linear(x, w, b))
            t_nn = timed(lambda: torch.addmm(b, x, wt, out=out))
            t_tn_nb = timed(lambda: F.linear(x, w))
            t_nn_nb = timed(lambda: torch.mm(x, wt, out=out))
            print(f"M={M} K={K} N={N}: linear(TN)+bias {t_tn:7.1f} us {fl / t_tn / 1e6:5.0f} TF | addmm(NN)+bias {t_nn:7.1f} us "
                  f"{fl / t_nn / 1e6:5.0f} TF | TN no bias {t_tn_nb:7.1f} | NN no bias {t_nn_nb:7.1f}", flush=True)


if __name__ == "__main__":
    sys.exit(main())
