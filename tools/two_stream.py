#!/usr/bin/env python3
"""Experiment: does running two half-batches of the headline forward on two HIP streams (the memory-bound merge /
LayerNorm / GELU passes of one half under the MFMA-bound GEMMs and attention of the other) beat one stream?

    python tools/two_stream.py [--batch 128] [--streams 2] [--steps 10]

Result on this pool (round 2): two forwards in flight on two streams of one process never finish -- also with
r=0 and every TOME_* kernel switched off (TOME_ATTN_KERNEL=0 TOME_FUSE_NEXT=0 TOME_GELU_KERNEL=0), i.e. with the
framework's kernels alone; each replica alone on its side stream is fine.  Round 3 reduced it to two chains of plain
torch.mm (tools/probes/two_stream_gemm.py: 88 ms on one stream; not finished after 20 s on two).  Every GEMM of the
model runs as a persistent Stream-K kernel (`..._SK3_..._MT256x256x64`, one workgroup per CU spinning on its siblings'
partial tiles) under either BLAS preference (profiles/r04_two_stream_probe_rocblas_kernel.txt), so two such grids
waiting on each other is the likely cause; the probe has no non-Stream-K control.  Since then a patched forward
issued while another is in flight on a different stream is ORDERED behind it (tome/patch/_common.py,
`_guard_one_forward_in_flight`): this script now finishes, and shows that two streams buy nothing.  The script still
gives up after 40 s instead of hanging, with the host stack dumped at 30 s.

Prints clips/s for one stream over the whole batch and for S streams over batch/S clips each (S replicas of the
patched model sharing nothing but the device; `_tome_info` is per model, so one forward at a time per replica).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--r", type=int, default=16)
    ap.add_argument("--offset", type=int, default=0, help="start replica k's forward k*offset layers late (host order)")
    a = ap.parse_args()
    import faulthandler
    faulthandler.dump_traceback_later(30, repeat=True)  # where the host is, should a step never come back
    import tome
    from hosts.videomae import videomae_base
    dev = torch.device("cuda", 0)
    S = a.streams
    models = []
    for _ in range(S):
        torch.manual_seed(0)
        m = videomae_base(num_frames=a.frames).to(dev).to(torch.bfloat16).eval()
        tome.patch.videomae(m, prop_attn=False)
        m.r = a.r
        models.append(m)
    clips = torch.rand(a.batch, 3, a.frames, 224, 224, device=dev).to(torch.bfloat16)
    parts = list(clips.chunk(S))
    streams = [torch.cuda.Stream() for _ in range(S)]

    def one():
        return models[0]([clips])

    def multi():
        cur = torch.cuda.current_stream()
        outs = []
        for s, m, c in zip(streams, models, parts):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(m([c]))
        for s in streams:
            cur.wait_stream(s)
        return outs

    def timed(fn, label):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{label}: {a.batch * a.steps / dt:.1f} clips/s ({dt / a.steps * 1e3:.2f} ms per step)", flush=True)

    print("models built", flush=True)
    with torch.no_grad():
        ref = one().float()
        torch.cuda.synchronize()
        print("one stream: done", flush=True)
        for s_, m_, c_ in zip(streams, models, parts):  # each replica alone on its stream first
            with torch.cuda.stream(s_):
                m_([c_])
            torch.cuda.synchronize()
            print("replica alone on its stream: done", flush=True)
        outs = multi()
        done = torch.cuda.Event()
        done.record()
        t_wait = time.perf_counter()
        while not done.query():
            if time.perf_counter() - t_wait > 40:
                print("the concurrent step has not finished after 40 s: giving up", flush=True)
                os._exit(3)
            time.sleep(0.05)
        got = torch.cat(outs).float()
        print("max |logit diff| one vs multi:", float((ref - got).abs().max()), flush=True)
    for _ in range(2):
        timed(one, f"1 stream  x {a.batch}")
        timed(multi, f"{S} streams x {a.batch // S}")


if __name__ == "__main__":
    main()
