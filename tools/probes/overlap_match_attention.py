#!/usr/bin/env python3
"""Probe: does the matching of a block (tome_match_keys: needs only the keys, which exist as soon as the qkv GEMM is
done) hide behind the block's attention when it runs on a second HIP stream?  The attention kernel is matrix/vector
bound with HBM nearly idle and leaves ~64 VGPRs per SIMD; k_unit_rows_f (28 VGPRs, HBM bound) is the matching's
largest piece.  One stream-K GEMM at a time only (two resident ones wait for each other: DESIGN_HISTORY).
    python3 tools/probes/overlap_match_attention.py [batch ...]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]
import torch  # noqa: E402

from tome import _abi  # noqa: E402

dev = torch.device("cuda", 0)
H, N, C = 12, 1568, 768
REP = 10


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REP):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REP * 1e3


for B in [int(a) for a in sys.argv[1:]] or [128, 384]:
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3, H, 64, device=dev).bfloat16()
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    w = (torch.randn(C, C, device=dev) * 0.02).bfloat16()
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    keep = {}

    def attn():
        keep["o"] = _abi.prop_attention(q, k, v, None, 0.125)

    def proj():
        keep["p"] = torch.nn.functional.linear(keep["o"].reshape(B * N, C), w)

    def match():
        keep["plan"] = _abi.match_keys(k, 16)

    def serial():
        attn()
        match()

    def serial_proj():
        attn()
        proj()
        match()

    def fork(first_side):
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        if first_side:
            with torch.cuda.stream(side):
                match()
            attn()
        else:
            attn()
            with torch.cuda.stream(side):
                match()

    def join():
        ev = torch.cuda.Event()
        ev.record(side)
        main.wait_event(ev)

    def overlapped(first_side=True):
        fork(first_side)
        join()

    def overlapped_proj(first_side=True):
        fork(first_side)
        proj()
        join()

    t_a, t_m = timed(attn), timed(match)
    attn()
    t_p = timed(proj)
    print(f"batch {B}: attention {t_a:8.1f} us   matching {t_m:7.1f} us   proj GEMM {t_p:7.1f} us", flush=True)
    t_s = timed(serial)
    t_o1 = timed(lambda: overlapped(True))
    t_o2 = timed(lambda: overlapped(False))
    print(f"  attention ; matching          one stream {t_s:8.1f} us   two streams, matching issued first {t_o1:8.1f}"
          f"   attention issued first {t_o2:8.1f}", flush=True)
    t_sp = timed(serial_proj)
    t_p1 = timed(lambda: overlapped_proj(True))
    t_p2 = timed(lambda: overlapped_proj(False))
    print(f"  attention ; proj ; matching   one stream {t_sp:8.1f} us   two streams, matching issued first {t_p1:8.1f}"
          f"   attention issued first {t_p2:8.1f}", flush=True)
    del qkv, q, k, v, keep
    torch.cuda.empty_cache()
