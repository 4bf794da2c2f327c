"""Where does the fused add+merge+LN kernel lose time?  Same shapes, different index patterns."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch
from tome import _abi, merge as tm

dev = torch.device("cuda", 0)
B, T, C = int(os.environ.get("EXP_B", "64")), 1568, 768
torch.manual_seed(0)
x = torch.randn(B, T, C, device=dev, dtype=torch.bfloat16)
a = torch.randn(B, T, C, device=dev, dtype=torch.bfloat16)
w = torch.ones(C, device=dev, dtype=torch.bfloat16)
b = torch.zeros(C, device=dev, dtype=torch.bfloat16)
size = torch.ones(B, T, 1, device=dev, dtype=torch.bfloat16)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(tag, r, sort_unm=False, sz=size, addend=a):
    metric = torch.randn(B, T, 64, device=dev)
    mg, _ = tm.bipartite_soft_matching(metric, r)
    plan = mg.plan
    if sort_unm:
        plan.unm_idx = plan.unm_idx.sort(dim=1)[0].contiguous()
    us = timeit(lambda: _abi.merge_wavg_ln(plan, x, sz, w, b, 1e-6, addend=addend))
    nbytes = B * ((2 if addend is not None else 1) * T * C * 2 + 2 * (T - r) * C * 2)
    us2 = timeit(lambda: _abi.merge_wavg(plan, x, sz))
    nb2 = B * (T * C * 2 + (T - r) * C * 2)
    print(f"{tag:34s} r={r:3d} add+merge+ln {us:7.1f} us {nbytes / us / 1e6:6.2f} TB/s | merge {us2:6.1f} us {nb2 / us2 / 1e6:5.2f} TB/s")


run("random unm order", 16)
run("sorted unm order", 16, sort_unm=True)
run("r=1 random", 1)
run("r=1 sorted", 1, sort_unm=True)
run("size=None random", 16, sz=None)
run("no addend random", 16, addend=None)
run("no addend sorted", 16, sort_unm=True, addend=None)
us = timeit(lambda: _abi.add_layernorm(x, a, w, b, 1e-6))
print(f"add_layernorm same rows {us:7.1f} us {B * T * C * 2 * 4 / us / 1e6:6.2f} TB/s")

ob = torch.randn(C, device=dev, dtype=torch.bfloat16)
metric = torch.randn(B, T, 64, device=dev)
plan = tm.bipartite_soft_matching(metric, 16)[0].plan
for tag, kw in (("out_bias=None", {}), ("out_bias", {"out_bias": ob})):
    us = timeit(lambda: _abi.merge_wavg_ln(plan, x, size, w, b, 1e-6, addend=a, **kw))
    print(f"add+merge+ln {tag:14s} {us:7.1f} us")
us = timeit(lambda: _abi.add_layernorm(x, None, w, b, 1e-6))
print(f"layernorm only (no addend)  {us:7.1f} us {B * T * C * 2 * 2 / us / 1e6:6.2f} TB/s")
y = torch.empty_like(x)
us = timeit(lambda: y.copy_(x))
print(f"torch copy                  {us:7.1f} us {B * T * C * 2 * 2 / us / 1e6:6.2f} TB/s")
us = timeit(lambda: torch.add(x, a, out=y))
print(f"torch add                   {us:7.1f} us {B * T * C * 2 * 3 / us / 1e6:6.2f} TB/s")
