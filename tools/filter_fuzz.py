#!/usr/bin/env python3
"""Randomised campaign: the candidate-filter path of the matching (csrc/tome_match_filter.h) against the fp32 pass, on
bf16 keys -- indices and node_max bits must be identical.  Shapes, key statistics (plain normal, per-token scales,
low-rank structure that packs cosines close together, duplicated tokens, norms at both ends of the filter's trusted
range) and r are drawn at random.
    python tools/filter_fuzz.py [cases] [seed]"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

from tome import _abi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), device=dev, generator=g).item())
bad = 0
t0 = time.time()
cand_hist = {}
for c in range(cases):
    T = [ri(2, 400), ri(300, 1600), 197, 784, 1568, 3137][ri(0, 5)]
    n = max(1, min(ri(1, 24), 40000 // T))
    heads = ri(0, 1) == 1
    D = 64 if heads else [8, 16, 24, 40, 64][ri(0, 4)]
    kind = ri(0, 6)
    shape = (n, T, 12, 64) if heads else (n, T, D)
    x = torch.randn(shape, device=dev, generator=g)
    if kind == 1:      # per-token scales over many binades
        x = x * torch.exp2(torch.randint(-20, 21, shape[:2] + (1,) * (len(shape) - 2), device=dev, generator=g).float())
    elif kind == 2:    # low rank + small noise: cosines packed close together
        k = ri(1, 3)
        basis = torch.randn((n, k) + shape[2:], device=dev, generator=g)
        coef = torch.randn((n, T, k), device=dev, generator=g)
        x = torch.einsum("ntk,nk...->nt...", coef, basis) + 10.0 ** (-ri(1, 4)) * x
    elif kind == 3:    # duplicated tokens (exact ties)
        rep = ri(2, 6)
        x = x[:, : max(1, (T + rep - 1) // rep)].repeat_interleave(rep, dim=1)[:, :T]
    elif kind == 4:    # nearly constant token + tiny differences (all cosines within 1e-4 of 1)
        x = x[:, :1] + 10.0 ** (-ri(2, 3)) * x
    elif kind == 5:    # norms 2^-70 .. 2^-40: products of two tokens' channels are denormal (FILT_NORM_LO = 1e-14)
        x = x * torch.exp2(torch.randint(-70, -39, shape[:2] + (1,) * (len(shape) - 2), device=dev, generator=g).float())
    elif kind == 6:    # norms 2^40 .. 2^62 around FILT_NORM_HI = 1e18, squared norms up to overflow
        x = x * torch.exp2(torch.randint(40, 63, shape[:2] + (1,) * (len(shape) - 2), device=dev, generator=g).float())
    x = x.bfloat16()
    r = [1, 5, 16, 32, T][ri(0, 4)]
    cls, dist = ri(0, 3) == 0, ri(0, 7) == 0
    if heads:
        qkv = torch.zeros(n, T, 3, 12, 64, device=dev, dtype=torch.bfloat16)
        qkv[:, :, 1] = x
        keys = qkv.permute(2, 0, 3, 1, 4)[1]
        call = lambda: _abi.match_keys(keys, r, cls, dist, want_node_max=True)
    else:
        call = lambda: _abi.match(x, r, cls, dist, want_node_max=True)
    os.environ["TOME_SCORES_FILTER"] = "0"
    want = call()
    os.environ["TOME_SCORES_FILTER"] = "2"
    got = call()
    if want is None:
        assert got is None
        continue
    same = all(torch.equal(getattr(got, k), getattr(want, k)) for k in ("src_idx", "dst_idx", "unm_idx")) and \
        torch.equal(got.node_max.view(torch.int32), want.node_max.view(torch.int32))
    if not same:
        bad += 1
        print(f"MISMATCH case {c}: n={n} T={T} D={D} heads={heads} kind={kind} r={r} cls={cls} distill={dist}", flush=True)
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.1f} s (seed {seed})")
sys.exit(1 if bad else 0)
