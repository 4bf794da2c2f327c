#!/usr/bin/env python3
"""Probe: do two HIP streams of one process make progress concurrently on this pool?  (tools/two_stream.py: two patched
forwards on two streams never finish.)  Stages: matmuls only; matmul + elementwise; with a 40 s bail-out each."""
import os
import sys
import time
import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev).bfloat16()
b = torch.randn(8192, 8192, device=dev).bfloat16()
x = torch.randn(64 * 1568, 3072, device=dev).bfloat16()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def wait(tag):
    done = torch.cuda.Event()
    done.record()
    t0 = time.perf_counter()
    while not done.query():
        if time.perf_counter() - t0 > 40:
            print(f"{tag}: NOT finished after 40 s", flush=True)
            os._exit(3)
        time.sleep(0.01)
    print(f"{tag}: finished in {time.perf_counter() - t0:.3f} s", flush=True)


def run(tag, f1, f2, reps=20):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    for _ in range(reps):
        with torch.cuda.stream(s1):
            f1()
        with torch.cuda.stream(s2):
            f2()
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    wait(tag)


torch.cuda.synchronize()
run("matmul | matmul", lambda: a @ b, lambda: b @ a)
run("matmul | gelu", lambda: a @ b, lambda: F.gelu(x))
lin = torch.nn.Linear(768, 3072).to(dev).bfloat16()
y = torch.randn(64 * 1568, 768, device=dev).bfloat16()
run("linear+bias | gelu", lambda: lin(y), lambda: F.gelu(x))
conv = torch.nn.Conv3d(3, 768, (2, 16, 16), (2, 16, 16)).to(dev).bfloat16()
clip = torch.rand(8, 3, 16, 224, 224, device=dev).bfloat16()
run("conv3d | gelu", lambda: conv(clip), lambda: F.gelu(x), reps=3)
run("sdpa | gelu", lambda: F.scaled_dot_product_attention(y.view(64, 1568, 12, 64).transpose(1, 2), y.view(64, 1568, 12, 64).transpose(1, 2), y.view(64, 1568, 12, 64).transpose(1, 2)), lambda: F.gelu(x), reps=5)
# timing: serial vs concurrent
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def serial():
    for _ in range(20):
        lin(y); F.gelu(x)
def conc():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for _ in range(20):
        with torch.cuda.stream(s1): lin(y)
        with torch.cuda.stream(s2): F.gelu(x)
    cur.wait_stream(s1); cur.wait_stream(s2)
serial(); conc()
print(f"20 x (linear, gelu): one stream {timed(serial):.2f} ms, two streams {timed(conc):.2f} ms", flush=True)
