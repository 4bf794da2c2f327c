#!/usr/bin/env python3
"""Headline benchmark: clips/s of a VideoMAE-B 16x224x224 forward with ToMe r=16 (BASELINE.json
configs[1]), bf16, synthetic clips resident in HBM, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one forward of the patched model over this rank's batch of clips (patch embed -> 12 blocks,
each with the HIP merge path between attention and MLP -> head -> top-1/top-5 counts on device).  Ranks
are independent replicas on different clips (weak scaling, no data-path collective); one RCCL
all-reduce of [top1, top5, clips] closes the timed region.  Rank 0 prints ONE JSON line.

Besides the throughput the line carries
  roofline      the dominant merge-path kernel against its bound (HBM 8 TB/s or fp32-MFMA 157.3 TF/s,
                /opt/skills/guides/MI355X_MICROARCH.md), from HIP-event timings taken in this run
  cpu_baseline  the reference's PyTorch-CPU path (restated in oracle/torch_port.py, kind "port") timed
                on this host's cores on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")
for _p in (ROOT, PKG_DIR):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak (same table)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same table); the patched attention runs on it

EMBED, HEADS, HEAD_DIM, LAYERS = 768, 12, 64, 12
DEFAULT_BATCH = 384  # clips per GPU per step: 64 -> 128 is +4.7 % clips/s (GEMM / attention efficiency), 128 -> 256 -> 384 -> 512: 2327 -> 2369 -> 2396 -> 2396


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH, help="clips per GPU per step")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--r", type=int, default=16)
    ap.add_argument("--cpu-clips", type=int, default=2, help="batch of the CPU baseline sample")
    ap.add_argument("--cpu-iters", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured HIP graph instead of launching its kernels from the host "
                         "(measured at batch 128: 2269-2273 vs 2265-2272 clips/s -- the device is never idle there)")
    ap.add_argument("--isolated", action="store_true",
                    help="also time the two streaming kernels back to back on resident inputs (roofline.back_to_back)")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads of the `also` object")
    ap.add_argument("--also-quick", action="store_true", help="`also` at an eighth of the batch and few iterations (tests)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "rehearse N>1 on a one-GPU box)")
    ap.add_argument("--force-group", action="store_true", help="form the torch.distributed group at N=1 too (a group of "
                    "one: RCCL initialisation and the job's collectives run on a single GPU; tests/test_launch_gpu.py)")
    return ap.parse_args()


def token_schedule(t0: int, r: int, layers: int):
    out, t = [], t0
    for _ in range(layers):
        re = max(0, min(r, t // 2))
        out.append((t, re))
        t -= re
    return out


class LaunchTimer:
    """HIP events around every launch of the two streaming entry points of the merge path WHILE THE PATCHED MODEL
    RUNS: tome_merge_wavg_ln (residual add + merge + norm2, `k_merge_rows`) and tome_add_layernorm (`k_add_ln_rows`).
    The kernels are launched on torch's current stream, the stream these events are recorded on.  Durations are those
    of the launches in their real context (behind the projection GEMM, with the matching's index buffers fresh), which
    is also what a rocprofv3 kernel trace of the same command shows; `measure_kernels(..., isolated=True)` times the
    same kernels back to back on resident inputs instead (5 % slower at batch 384: 650 vs 616 us for identical launches
    in the trace)."""

    def __init__(self):
        self.rec = {"k_merge_rows": [], "k_add_ln_rows": []}
        self._saved = {}

    def _wrap(self, name, fn, nbytes):
        def timed(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.rec[name].append((e0, e1) + nbytes(*a, **kw))
            return out
        return timed

    @staticmethod
    def _merge_bytes(plan, x, size, weight, bias, eps, addend=None, log_size=False, out_bias=None):
        n, t, c = x.shape
        e, es, to = x.element_size(), 2, t - plan.r
        fused = n * (t * c * e * (2 if addend is not None else 1) + 2 * to * c * e + (t * es if size is not None else 0) + to * es)
        d8 = n * (t * c * e + (t * es if size is not None else 0) + to * c * e + to * es)  # x, size in; x', size' out
        return fused, d8

    @staticmethod
    def _regrouped_bytes(plan, x_full, size, frames, has_cls=True, ln=None, addend=None, log_size=False,
                         addend_grouped=None, cls_addend=None, out_bias=None):
        b, n_tok, c = x_full.shape
        e, es, cls, p, f, r = x_full.element_size(), 2, 1 if has_cls else 0, plan.T, int(frames), plan.r
        rows_in, rows_out = b * n_tok, b * (cls + (p - r) * f)
        fused = rows_in * c * e + rows_out * c * e * (2 if ln is not None else 1)
        if addend is not None:
            fused += rows_in * c * e
        if addend_grouped is not None:  # the spatial attention's output where it lies: P rows per group + the class addend
            fused += plan.n * p * c * e + (b * c * e if cls_addend is not None else 0)
        fused += (plan.n * p * es if size is not None else 0) + plan.n * (p - r) * es
        d8 = plan.n * (p * c * e + (p * es if size is not None else 0) + (p - r) * c * e + (p - r) * es)
        return fused, d8

    @staticmethod
    def _add_ln_bytes(x, addend, weight, bias, eps, skip_first=False):
        rows, c, e = x.numel() // x.shape[-1], x.shape[-1], x.element_size()
        return rows * c * e * (4 if addend is not None else 2), 0  # read x (, a); (write x',) write y

    def __enter__(self):
        from tome import _abi
        self._abi = _abi
        for attr, name, nb in (("merge_wavg_ln", "k_merge_rows", self._merge_bytes),
                               ("merge_wavg_regrouped", "k_merge_rows", self._regrouped_bytes),
                               ("add_layernorm", "k_add_ln_rows", self._add_ln_bytes)):
            self._saved[attr] = getattr(_abi, attr)
            setattr(_abi, attr, self._wrap(name, self._saved[attr], nb))
        return self

    def __exit__(self, *exc):
        for attr, fn in self._saved.items():
            setattr(self._abi, attr, fn)
        return False

    def stats(self, steps: int):
        """Per kernel: milliseconds, bytes and launches per step (mean over the instrumented steps)."""
        torch.cuda.synchronize()
        out = {}
        for name, rec in self.rec.items():
            ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in rec)
            out[name] = {"ms": ms / steps, "bytes": sum(b for _, _, b, _ in rec) // steps, "flops": 0,
                         "launches": len(rec) // steps, "bytes_8d": sum(d for _, _, _, d in rec) // steps}
        return out


def measure_kernels(batch: int, t0: int, r: int, dev, reps: int = 10, isolated: bool = True):
    """HIP-event timing of the merge-path kernels at the 12 layer shapes of this workload, on the stream
    they are launched on (torch's current stream).  Returns per-kernel totals over one forward's launches.
    isolated=False: only the three matching stages (timed inside tome_match_keys); the two streaming kernels are then
    timed inside the forward by LaunchTimer."""
    import ctypes
    from tome import _abi
    L = _abi.lib()
    # the stage events live in the MEASUREMENT build of the library (csrc/build.py builds lib/libtome_hip_prof.so
    # with -DTOME_PROFILE_HOOKS beside the product library; the product ABI has no such entry points).  It is bound
    # here, swapped in for the matching calls of this function only, and never touches the forward that is timed.
    prof_path = os.path.join(os.path.dirname(_abi.LIB_PATH), "libtome_hip_prof.so")
    P = _abi.bind(prof_path)
    P.tome_profile_enable.restype = ctypes.c_int
    P.tome_profile_enable.argtypes = [ctypes.c_int]
    P.tome_profile_read.restype = ctypes.c_int
    P.tome_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_int]

    def profile_read():
        buf = (ctypes.c_float * 3)()
        assert P.tome_profile_read(buf, 3) == 0
        return [float(v) for v in buf]

    def profiled_match(keys, re, reps):
        acc, plan = [0.0, 0.0, 0.0], None
        _abi._lib = P
        try:
            assert P.tome_profile_enable(reps) == 0  # every stage kernel is launched `reps` times between events
            for i in range(3):
                plan = _abi.match_keys(keys, re)
                ms = profile_read()
                if i >= 1:
                    acc = [a + b / 2 for a, b in zip(acc, ms)]
            assert P.tome_profile_enable(0) == 0
        finally:
            _abi._lib = L
        return plan, acc
    sched = [(t, re) for t, re in token_schedule(t0, r, LAYERS) if re > 0]
    g = torch.Generator(device=dev).manual_seed(7)
    stats = {
        "k_unit_rows_heads": {"ms": 0.0, "bytes": 0, "flops": 0, "launches": 0},
        "k_scores_rowmax": {"ms": 0.0, "bytes": 0, "flops": 0, "launches": 0},
        "k_rank_select": {"ms": 0.0, "bytes": 0, "flops": 0, "launches": 0},
        "k_merge_rows": {"ms": 0.0, "bytes": 0, "flops": 0, "launches": 0},
        "k_add_ln_rows": {"ms": 0.0, "bytes": 0, "flops": 0, "launches": 0},
    }
    st = torch.cuda.current_stream(dev).cuda_stream
    # the tokens and their sizes evolve through the 12 layers exactly as in a forward (most sizes stay 1,
    # merged tokens carry 2, 3, ...); the keys of every layer are fresh random bf16 tensors
    x = torch.randn(batch, t0, EMBED, device=dev, generator=g).bfloat16()
    ln_w = (1.0 + 0.1 * torch.randn(EMBED, device=dev, generator=g)).bfloat16()
    ln_b = (0.1 * torch.randn(EMBED, device=dev, generator=g)).bfloat16()
    size = None
    for t, re in sched:
        assert x.shape[1] == t
        # per-head keys as the attention leaves them: the k slice of a [B, T, 3, H, 64] qkv buffer
        qkv = torch.randn(batch, t, 3, HEADS, HEAD_DIM, device=dev, generator=g).bfloat16()
        keys = qkv.permute(2, 0, 3, 1, 4)[1]
        # --- matching: per-stage events recorded inside tome_match_keys
        plan, acc = profiled_match(keys, re, reps)
        t1, t2 = (t + 1) // 2, t // 2
        for name, ms in zip(("k_unit_rows_heads", "k_scores_rowmax", "k_rank_select"), acc):
            stats[name]["ms"] += ms
            stats[name]["launches"] += 1
        # read 12 heads; write the bf16 head mean in fragment order + norm + 1/norm (the candidate-filter path of the
        # matching, csrc/tome_match_filter.h; round 2 wrote fp32 unit vectors: 4 bytes per channel instead of 2)
        stats["k_unit_rows_heads"]["bytes"] += batch * t * (HEAD_DIM * (2 * HEADS + 2) + 8)
        stats["k_scores_rowmax"]["flops"] += batch * 2 * t1 * t2 * HEAD_DIM
        stats["k_scores_rowmax"]["bytes"] += batch * t * HEAD_DIM * 4
        stats["k_rank_select"]["bytes"] += batch * (t1 * 8 + t1 * 8)
        # --- residual add + merge + the block's norm2: one kernel per call (tome_merge_wavg_ln with addend, what
        # the patched block runs), timed over back-to-back launches with preallocated outputs
        res = (0.1 * torch.randn(batch, t, EMBED, device=dev, generator=g)).bfloat16()
        x_out = torch.empty(batch, t - re, EMBED, device=dev, dtype=torch.bfloat16)
        y_out = torch.empty_like(x_out)
        s_out = torch.empty(batch, t - re, 1, device=dev, dtype=torch.bfloat16)
        sp = None if size is None else size.data_ptr()

        def launch():
            rc = L.tome_merge_wavg_ln(x.data_ptr(), 1, sp, 1, batch, t, EMBED, re, plan.src_idx.data_ptr(),
                                      plan.dst_idx.data_ptr(), plan.unm_idx.data_ptr(), 0, None, ln_w.data_ptr(),
                                      ln_b.data_ptr(), 1e-6, res.data_ptr(), x_out.data_ptr(), y_out.data_ptr(),
                                      s_out.data_ptr(), None, None, st)
            assert rc == 0
        launch()  # (always once: the next layer's tokens and sizes come from it)
        if not isolated:
            x, size = x_out, s_out
            continue
        launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        e1.synchronize()
        stats["k_merge_rows"]["ms"] += e0.elapsed_time(e1) / reps
        stats["k_merge_rows"]["launches"] += 1
        # SURVEY 8d (read x and size once, write x' and size' once; bf16 tokens and sizes) + the fused residual's
        # second input and the fused norm2's own output y = LayerNorm(x') (its input never leaves the kernel)
        stats["k_merge_rows"]["bytes"] += batch * (2 * t * EMBED * 2 + t * 2 + 2 * (t - re) * EMBED * 2 + (t - re) * 2)
        stats["k_merge_rows"]["bytes_8d"] = stats["k_merge_rows"].get("bytes_8d", 0) + \
            batch * (t * EMBED * 2 + t * 2 + (t - re) * EMBED * 2 + (t - re) * 2)
        # --- second residual + next block's norm1 (tome_add_layernorm) on the merged tokens
        if len(stats["k_add_ln_rows"]) and t != sched[-1][0]:
            x2 = torch.empty_like(x_out)

            def launch2():
                rc = L.tome_add_layernorm(x_out.data_ptr(), y_out.data_ptr(), 1, batch * (t - re), EMBED, ln_w.data_ptr(),
                                          ln_b.data_ptr(), 1e-6, x2.data_ptr(), y_out.data_ptr(), st)
                assert rc == 0
            for _ in range(2):
                launch2()
            e0.record()
            for _ in range(reps):
                launch2()
            e1.record()
            e1.synchronize()
            stats["k_add_ln_rows"]["ms"] += e0.elapsed_time(e1) / reps
            stats["k_add_ln_rows"]["launches"] += 1
            stats["k_add_ln_rows"]["bytes"] += batch * (t - re) * EMBED * 2 * 4  # read x, a; write x', y
        x, size = x_out, s_out
    return stats


def measure_attention(batch: int, t0: int, r: int, dev, reps: int = 5):
    """The patched blocks' attention (tome_prop_attention, the caller on the near side of the merge path) over
    the token counts of the 12 layers: device time per forward and TFLOP/s against the dense bf16 MFMA peak.
    Reported next to the merge path, not as part of `roofline` (SURVEY.md 8d prices the merge path)."""
    from tome import _abi
    g = torch.Generator(device=dev).manual_seed(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = flops = 0.0
    for t, _ in token_schedule(t0, r, LAYERS):
        qkv = torch.randn(batch, t, 3, HEADS, HEAD_DIM, device=dev, generator=g).bfloat16()
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        for _ in range(2):
            _abi.prop_attention(q, k, v, None, HEAD_DIM ** -0.5)
        e0.record()
        for _ in range(reps):
            _abi.prop_attention(q, k, v, None, HEAD_DIM ** -0.5)
        e1.record()
        e1.synchronize()
        ms += e0.elapsed_time(e1) / reps
        flops += 4.0 * batch * HEADS * t * t * HEAD_DIM
        del qkv, q, k, v
    tf = flops / (ms / 1e3) / 1e12
    return {"kernel": "k_prop_attention", "ms_per_step": round(ms, 3), "launches_per_step": LAYERS,
            "TFLOP/s": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4)}


def _eff_r(t: int, r: int, cls: bool) -> int:
    return max(0, min(r, (t - (1 if cls else 0)) // 2))


def measure_merge_kernel(dev, batch: int, frames: int, tokens: int, r: int, cls: bool, reps: int = 5):
    """The merge kernel of one model family (residual add + size-weighted merge + LayerNorm, the launch the patched
    block makes) over the token counts of its 12 layers, HIP events around back-to-back launches on torch's current
    stream.  frames == 1: one contiguous sequence per clip (VideoMAE; ViViT with its class token, cls=True);
    frames > 1: `frames` interleaved groups of `tokens` spatial tokens behind a class token (TimeSformer,
    Motionformer: tome_merge_wavg_regrouped_ln).  Returns per-forward totals: milliseconds, launches, the bytes the
    fused launch moves (read x and the residual, write x' and LayerNorm(x'), sizes) and the merge-only bytes of
    SURVEY.md 8d (read x and sizes once, write x' and sizes' once)."""
    from tome import _abi
    g = torch.Generator(device=dev).manual_seed(11)
    n = batch * frames
    t = tokens
    ncls = 1 if frames > 1 else 0
    x = torch.randn(batch, ncls + t * frames, EMBED, device=dev, generator=g).bfloat16()
    ln_w = (1.0 + 0.1 * torch.randn(EMBED, device=dev, generator=g)).bfloat16()
    ln_b = (0.1 * torch.randn(EMBED, device=dev, generator=g)).bfloat16()
    size = None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = {"ms": 0.0, "launches": 0, "bytes_fused": 0, "bytes_8d": 0}
    for _ in range(LAYERS):
        re = _eff_r(t, r, cls)
        if re <= 0:
            break
        keys = torch.randn(n, HEADS, t, HEAD_DIM, device=dev, generator=g).bfloat16()
        plan = _abi.match_keys(keys, re, cls)
        res = (0.1 * torch.randn(x.shape, device=dev, generator=g)).bfloat16()
        if frames > 1:
            def launch():
                return _abi.merge_wavg_regrouped(plan, x, size, frames, has_cls=True, ln=(ln_w, ln_b, 1e-6), addend=res)
        else:
            def launch():
                return _abi.merge_wavg_ln(plan, x, size, ln_w, ln_b, 1e-6, addend=res)
        for _ in range(2):
            out = launch()
        e0.record()
        for _ in range(reps):
            out = launch()
        e1.record()
        e1.synchronize()
        rows_in, rows_out = batch * (ncls + t * frames), batch * (ncls + (t - re) * frames)
        tot["ms"] += e0.elapsed_time(e1) / reps
        tot["launches"] += 1
        tot["bytes_fused"] += 2 * rows_in * EMBED * 2 + 2 * rows_out * EMBED * 2 + n * t * 2 + n * (t - re) * 2
        tot["bytes_8d"] += n * (t * EMBED * 2 + t * 2 + (t - re) * EMBED * 2 + (t - re) * 2)
        x, size = out[0], out[2]
        t -= re
    return tot


def merge_roofline(tot, name: str):
    """`roofline`-shaped object of a merge kernel measured by measure_merge_kernel: achieved GB/s on the bytes the
    fused launch moves, and on the merge-only bytes of SURVEY.md 8d next to it."""
    sec = tot["ms"] / 1e3
    ach, ach8 = tot["bytes_fused"] / sec / 1e9, tot["bytes_8d"] / sec / 1e9
    return {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launches_per_step": tot["launches"],
            "avg_launch_us": round(tot["ms"] * 1e3 / max(1, tot["launches"]), 2),
            "bytes_fused": tot["bytes_fused"], "bytes_8d": tot["bytes_8d"], "achieved_8d": round(ach8, 1),
            "frac_8d": round(ach8 / HBM_PEAK_GBS, 4)}


def _throughput(model, clips, steps: int, warmup: int):
    with torch.no_grad():
        for _ in range(warmup):
            model([clips])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model([clips])
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"clips_per_s": round(clips.shape[0] / dt, 1), "ms_per_step": round(dt * 1e3, 3), "batch": clips.shape[0],
            "steps": steps, "warmup": warmup}


ALSO_FAMILIES = {
    # name: (host builder, patch name, frames of the clip, merge groups per clip, tokens per group, class token in the
    #        matching, yaml of the reference-style command line)
    "videomae_b_16x224": ("videomae", 16, 1, 1568, False, "configs/videomae_b_16x224.yaml"),
    "videomae_b_8x224": ("videomae", 8, 1, 784, False, None),
    "timesformer_divst_8x224": ("timesformer", 8, 8, 196, False, "configs/timesformer_divst_8x224.yaml"),
    "vivit_b_32x224": ("vivit", 32, 1, 3137, True, "configs/vivit_b_32x224.yaml"),
    "motionformer_224_16x4": ("motionformer", 16, 8, 196, False, "configs/motionformer_224_16x4.yaml"),
}


def _build_family(host: str, frames: int, dev):
    import tome
    from hosts import motionformer, timesformer, videomae, vivit
    torch.manual_seed(0)
    if host == "videomae":
        model, patch, kw = videomae.videomae_base(num_frames=frames), tome.patch.videomae, {"prop_attn": False}
    elif host == "timesformer":
        model, patch, kw = timesformer.timesformer_base(num_frames=frames), tome.patch.timesformer, {}
    elif host == "vivit":
        model, patch, kw = vivit.vivit_base(num_frames=frames), tome.patch.vivit, {}
    else:
        model, patch, kw = motionformer.motionformer_base(), tome.patch.motionformer, {}
    model = model.to(dev).to(torch.bfloat16).eval()
    patch(model, **kw)  # the reference's defaults: proportional attention on everywhere but VideoMAE
    return model


def also_workloads(dev, quick: bool = False):
    """The other sizes BASELINE.json / north_star name, in the same process and JSON line as the headline so that a
    driver-run record carries them: VideoMAE-B 8x224 r=16; TimeSformer divST 8x224 r = 8 / 16 / 32 (configs[2]);
    ViViT-B 32x224 (3137 tokens) r=64 (configs[3]); Motionformer 224 16x4 r=16 (configs[4], one GPU's share) -- bf16,
    random init, synthetic clips resident in HBM, each with the roofline of its family's merge kernel; and the
    reference's own harness protocol (slowfast/utils/model_benchmark.py:20-58: batch 8, fresh torch.rand clip per
    iteration, one event pair per forward, 5 warm-up + 100 timed) eager and replayed from a HIP graph."""
    from hosts import harness
    steps, warm = (3, 2) if quick else (12, 6)  # (warm-up also covers the clock ramp after the idle model build)
    plan = [  # (key, family, r values, clips per step)
        ("videomae_b_8x224", "videomae_b_8x224", (16,), 128),
        # (clips per step chosen like the headline's 384: where clips/s stops growing.  TimeSformer 64 -> 128 -> 192 ->
        #  256 -> 384: r=8 2452 -> 2582 -> 2622 -> 2666 -> 2684, r=16 3413 -> 3641 -> 3716 -> 3791 -> 3863, r=32 5536 ->
        #  6107 -> 6294 -> 6510 -> 6704; Motionformer 64 -> 96 -> 128: 2130 -> 2163 -> 2168 -- profiles/r04_batch_scan_*.txt)
        ("timesformer_divst_8x224", "timesformer_divst_8x224", (8, 16, 32), 384),
        ("vivit_b_32x224", "vivit_b_32x224", (64,), 64),   # (16 -> 64 clips per step: +11 %, GEMM efficiency)
        ("motionformer_224_16x4", "motionformer_224_16x4", (16,), 128),
    ]
    out = {}
    for key, fam, rs, batch in plan:
        host, frames, groups, tokens, cls, _ = ALSO_FAMILIES[fam]
        if quick:
            batch = max(2, batch // 8)
        model = _build_family(host, frames, dev)
        clips = torch.rand(batch, 3, frames, 224, 224, device=dev).to(torch.bfloat16)
        entry = {"workload": f"{fam}, bf16, random init, synthetic clips, {batch} clips per step", "tokens": tokens,
                 "merge_groups_per_clip": groups}
        if host == "vivit":
            entry["parity"] = ("block-level parity UNPINNED: the reference's tome/patch/vivit.py needs HF VivitSelfAttention, "
                               "absent from the installed transformers, and the reference holds no fixture; the merge "
                               "call itself (3137 tokens, class token) is pinned like every other")
        for r in rs:
            model.r = r
            rec = _throughput(model, clips, steps, warm)
            kname = "k_merge_rows_fast<LN> (regrouped layout)" if groups > 1 else "k_merge_rows_fast<LN>"
            # the family's merge kernel back to back on resident inputs (as in rounds 1-2), and inside two more forwards
            # (events around each launch)
            with torch.no_grad(), LaunchTimer() as lt:
                for _ in range(2):
                    model([clips])
                inf = lt.stats(2)["k_merge_rows"]
            with torch.no_grad():
                tot = measure_merge_kernel(dev, batch, groups, tokens, r, cls)
            rec["roofline"] = merge_roofline(tot, kname)
            rec["roofline"]["timed"] = "back-to-back launches on resident inputs (these launches are 40-200 us: an event pair " \
                                       "around each launch inside the forward adds its own few microseconds)"
            if inf["launches"] > 0 and inf["ms"] > 0:
                inside = merge_roofline({"ms": inf["ms"], "launches": inf["launches"], "bytes_fused": inf["bytes"],
                                         "bytes_8d": inf["bytes_8d"]}, kname)
                rec["roofline"]["in_forward"] = {"avg_launch_us": inside["avg_launch_us"], "frac": inside["frac"],
                                                 "launches_per_step": inside["launches_per_step"]}
            entry[f"r{r}"] = rec
        out[key] = entry
        del model, clips
        torch.cuda.empty_cache()
    # the reference's harness protocol and command line, batch 8
    proto = {}
    iters, wu = (10, 2) if quick else (100, 5)
    for fam, r in (("videomae_b_16x224", 16), ("timesformer_divst_8x224", 16), ("vivit_b_32x224", 64),
                   ("motionformer_224_16x4", 16)):
        host, frames, _, _, _, yaml_path = ALSO_FAMILIES[fam]
        row = {}
        for label, r_val, graph in (("r0", 0, False), (f"r{r}_eager", r, False), (f"r{r}_hip_graph", r, True)):
            opts = ["TRAIN.ENABLE", "False", "TEST.BATCH_SIZE", "8", "MODEL_BENCHMARK.WARMUP_ITERATIONS", str(wu),
                    "MODEL_BENCHMARK.ITERATIONS", str(iters), "TOME.ENABLE", str(r_val > 0), "TOME.R_VALUE", str(r_val)]
            cfg = harness.load_cfg(os.path.join(ROOT, yaml_path), opts)
            torch.manual_seed(cfg.RNG_SEED)
            model = harness.build_model(cfg).to(dev).to(torch.bfloat16).eval()
            harness.apply_tome(model, cfg, with_threshold=False)
            try:
                res = harness.perform_benchmark(model, cfg, dev, "bf16", 1, graph=graph)
                row[label] = round(res["clips_per_s"], 1)
            except Exception as exc:  # a host op that cannot be captured: say so instead of hiding the row
                row[label] = None
                row[label + "_error"] = f"{type(exc).__name__}: {exc}"[:200]
            del model
            torch.cuda.empty_cache()
        proto[fam] = row
    out["reference_protocol_batch8"] = dict(
        proto, unit="clips/s", protocol=f"slowfast/utils/model_benchmark.py:20-58 ({wu} warm-up + {iters} timed forwards, "
        "batch 8, fresh torch.rand clip per iteration, one event pair per forward), tools/model_benchmark.py command "
        "line; hip_graph = hosts/graphed.py replay")
    return out


def matching_placement(batch: int, tokens: int) -> str:
    """Where the timed forward runs the matching launches (tome/_overlap.py decides per layer)."""
    from tome import _overlap
    if _overlap.ENABLED and batch * tokens * tokens >= _overlap.MIN_WORK:
        return ("on a second HIP stream behind the layer's qkv GEMM, beside its attention and projection "
                "(tome/_overlap.py; profiles/r04_overlap_probe.txt): off the step's critical path -- the stage times "
                "of this object are standalone launches on an idle device")
    return "on the forward's stream, between projection and merge"


def roofline_of(stats, batch: int):
    """Roofline object of the kernel that takes the most device time per forward."""
    name = max(stats, key=lambda k: stats[k]["ms"])
    s = stats[name]
    sec = s["ms"] / 1e3
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # PMC bytes are per launch at the batch they were collected on: only quoted for that batch.  NOT measured
            # in this run: a stored result of the counter passes of tools/traffic_pmc.sh (rocprofv3 --pmc cannot run
            # inside the benchmark process), said so in `traffic_source`
            if tj.get("_batch", 64) == batch and tj.get(name) is not None:
                traffic = tj.get(name)
                traffic_source = (f"profiles/traffic.json: stored PMC passes (tools/traffic_pmc.sh, batch {batch}, "
                                  f"{tj.get('_collected', 'round 3')}), not measured in this run")
        except Exception:
            traffic = None
    if name == "k_scores_rowmax":
        achieved = s["flops"] / sec / 1e12
        return {"kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_source, "launches_per_step": s["launches"], "avg_launch_us": round(s["ms"] * 1e3 / s["launches"], 2)}
    achieved = s["bytes"] / sec / 1e9
    out = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
           "launches_per_step": s["launches"],
           "avg_launch_us": round(s["ms"] * 1e3 / s["launches"], 2), "bytes_fused": s["bytes"]}
    if s.get("bytes_8d"):
        # the same launches priced on the merge-only bytes of SURVEY.md 8d (the fused launch also reads the residual
        # and writes the LayerNorm output, which that formula does not count)
        out.update(bytes_8d=s["bytes_8d"], achieved_8d=round(s["bytes_8d"] / sec / 1e9, 1),
                   frac_8d=round(s["bytes_8d"] / sec / 1e9 / HBM_PEAK_GBS, 4))
    return out


def usable_cpus() -> int:
    """Host cores this process may actually use: the scheduler affinity mask, capped by the cgroup CPU
    quota when there is one (a GPU box hands each job a share of the host, not all of it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    cap = int(os.environ.get("TOME_BENCH_CPU_THREADS", "0"))
    if cap > 0:
        n = min(n, cap)
    return max(1, min(n, 64))


def cpu_baseline(frames: int, r: int, clips: int, iters: int):
    """The reference's PyTorch-CPU path (fp32) on a bounded sample: `iters` forwards of `clips` clips."""
    from hosts.videomae import videomae_base
    from oracle import torch_port
    torch.manual_seed(0)
    threads = usable_cpus()
    torch.set_num_threads(threads)
    host = videomae_base(num_frames=frames).eval()
    x = torch.rand(clips, 3, frames, 224, 224)
    torch_port.videomae_forward(host, x, r)  # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        torch_port.videomae_forward(host, x, r)
    dt = time.perf_counter() - t0
    return {"value": round(clips * iters / dt, 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{iters} forwards x {clips} clips, fp32, torch {torch.__version__} CPU, "
                      f"oracle/torch_port.py (op sequence of tome/merge.py + tome/patch/videomae.py), {dt:.1f}s"}


def worker(args):
    """One rank of the job (the whole job at N=1): everything that touches the GPU happens in here."""
    from hosts import launch
    rank, local_rank, world = launch.check_world(args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path in the product)")
    launch.require_one_gpu_per_rank(args.backend, world)  # (RCCL with more ranks than GPUs dies inside the library)
    if world > 1:
        # N ranks share the node's host cores: every rank keeps its share for the framework's intra-op pool (the
        # forward issues from one thread; an N x cores oversubscription only adds scheduling noise to the launch path)
        torch.set_num_threads(max(1, usable_cpus() // world))
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # "nccl" is RCCL on ROCm; a no-op at N=1 unless --force-group (a group of one: RCCL initialised, the collectives run)
    launch.init_process_group(args.backend, dev, force=args.force_group)
    grouped = dist.is_available() and dist.is_initialized()

    import tome
    from tome import _abi
    from hosts.videomae import videomae_base
    _abi.lib()  # fail loudly before timing anything if the HIP extension is missing

    torch.manual_seed(0)
    model = videomae_base(num_frames=args.frames).to(dev).to(torch.bfloat16).eval()
    tome.patch.videomae(model, prop_attn=False)
    model.r = args.r
    t0_tokens = model.model.patch_embed.num_patches

    B = args.batch
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    clips = torch.rand(B, 3, args.frames, 224, 224, device=dev, generator=gen).to(torch.bfloat16)
    labels = torch.randint(0, 400, (B,), device=dev, generator=gen)
    counts = torch.zeros(3, dtype=torch.int64, device=dev)  # top1, top5, clips

    from hosts.evalloop import all_reduce_counts, topk_counts

    def step():
        counts.add_(topk_counts(model([clips]), labels))

    graphed = False
    with torch.no_grad():
        step()  # setup, not a benchmark step: first-call work of the libraries (MIOpen solver search for the
        #         tubelet convolution, hipBLASLt heuristics, allocator growth) must not land in a timed step
        torch.cuda.synchronize()
        if args.graph:
            # The step (forward + top-k counts) has static shapes and launches everything on torch's current stream,
            # merge path included: captured once into a HIP graph, a step is one graph launch -- every kernel still
            # runs, the host-side launch gaps between the ~250 kernels of a forward do not (hosts/graphed.py does the
            # same for the reference-protocol rows).  Untimed, like the setup step.
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    step()
                torch.cuda.current_stream().wait_stream(side)
                hip_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(hip_graph):
                    step()
                eager_step = step

                def step():
                    hip_graph.replay()
                graphed = True
            except Exception as exc:  # capture refused: run eagerly and say so
                print(f"bench.py: HIP graph capture failed ({type(exc).__name__}: {exc}); eager steps", file=sys.stderr)
                step = eager_step if "eager_step" in dir() else step
                torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
        counts.zero_()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        for _ in range(args.steps):
            step()
        all_reduce_counts(counts)  # the one collective: top-1 / top-5 / clip counts over xGMI (no-op without a group)
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        elapsed = time.perf_counter() - t_start
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if grouped:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    census = launch.census(dev)  # ranks that answered + the device index of each (collective: every rank calls it)
    launch.check_census(census, args.backend, world)  # all ranks answered; under RCCL on distinct devices
    total_clips = int(counts[2].item())
    assert total_clips == B * args.steps * world, (total_clips, B, args.steps, world)

    out = None
    if rank == 0:
        out = {
            "metric": "clips/sec at r=16, VideoMAE-B 16x224x224 (forward with ToMe merge; merge indices bit-exact against the "
                      "fp32-arithmetic oracle, which the reference's fp32 fixtures pin)",
            "value": round(total_clips / elapsed, 2),
            "unit": "clips/s",
            "n_gpus": world,
            "ranks_seen": census["ranks_seen"],
            "rank_devices": census["devices"],
            "backend": ("rccl" if args.backend == "nccl" else args.backend) if grouped else None,
            "process_group": bool(grouped),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": f"VideoMAE-B random-init, synthetic {args.frames}x224x224 clips, r={args.r}, bf16 "
                            f"(BASELINE.json configs[1])",
                "clips_per_gpu_per_step": B, "tokens": t0_tokens, "r": args.r,
                "tokens_after_12_layers": token_schedule(t0_tokens, args.r, LAYERS)[-1][0]
                - token_schedule(t0_tokens, args.r, LAYERS)[-1][1],
                "parallelism": f"dp{world}: independent replicas, one RCCL all-reduce of [top1, top5, clips]",
                "step_launch": "HIP graph replay (forward + top-k counts captured once)" if graphed else "eager",
            },
            "top1": int(counts[0].item()), "top5": int(counts[1].item()),
        }
        if args.r <= 0:
            out["roofline"] = None  # r = 0: the unmerged model, no merge-path kernel runs
        elif not args.no_roofline:
            # the two streaming kernels: events around their launches inside two more forwards (untimed steps of the
            # same workload, after the timed region); the matching stages: timed inside tome_match_keys on the 12
            # layer shapes
            n_inst = 2
            with torch.no_grad(), LaunchTimer() as lt:
                for _ in range(n_inst):
                    (eager_step if graphed else step)()
                in_forward = lt.stats(n_inst)
            with torch.no_grad():
                stats = measure_kernels(B, t0_tokens, args.r, dev, isolated=args.isolated)
            back_to_back = {k: dict(stats[k]) for k in in_forward} if args.isolated else None
            stats.update(in_forward)
            out["roofline"] = roofline_of(stats, B)
            out["roofline"]["timed"] = (f"HIP events around each launch inside the forward ({n_inst} untimed steps after "
                                        "the timed region)")
            if back_to_back is not None and out["roofline"]["kernel"] in back_to_back:
                bb = back_to_back[out["roofline"]["kernel"]]
                out["roofline"]["back_to_back"] = {
                    "avg_launch_us": round(bb["ms"] * 1e3 / max(1, bb["launches"]), 2),
                    "frac": round(bb["bytes"] / (bb["ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if bb["ms"] > 0 else None}
            out["merge_path_kernels"] = {
                k: {"ms_per_step": round(v["ms"], 4), "launches": v["launches"],
                    "GB/s": round(v["bytes"] / (v["ms"] / 1e3) / 1e9, 1) if v["ms"] > 0 else None,
                    "TFLOP/s": round(v["flops"] / (v["ms"] / 1e3) / 1e12, 2) if v["flops"] else None}
                for k, v in stats.items()}
            out["merge_path_kernels"]["k_scores_rowmax"]["stage"] = (
                "similarity + row max of the matching, as one timed stage: k_scores_filter (approximate scores on the "
                "bf16 matrix pipe, candidate columns per row) + k_exact_rows (the contract's fp32 chain for the "
                "candidates; its fallback waves take the fp32 pass of tiles whose candidate lists overflowed -- none on "
                "this data); TFLOP/s = SURVEY 8d's 2*T1*T2*D per group / stage time, i.e. what an all-pairs fp32 pass "
                "would have to sustain (peak of that pipe: 157.3)")
            out["merge_path_ms_per_step"] = round(sum(v["ms"] for v in stats.values()), 4)
            # the PATH as SURVEY 8d prices it: per merge call read metric, x, size once and write x', size' once -- summed
            # over the 12 layers -- against the time of every launch of matching + merge (the second residual's
            # k_add_ln_rows is a block-wrapper pass, not part of 8d's path, and is left out of both sides)
            sched = [(t, re) for t, re in token_schedule(t0_tokens, args.r, LAYERS) if re > 0]
            bytes_8d = sum(B * (t * HEAD_DIM * 2 + t * EMBED * 2 + t * 2 + (t - re) * EMBED * 2 + (t - re) * 2)
                           for t, re in sched)
            path_ms = sum(stats[k]["ms"] for k in ("k_unit_rows_heads", "k_scores_rowmax", "k_rank_select", "k_merge_rows"))
            filt = B * ((t0_tokens + 1) // 2 + 31) // 32 >= 1024
            out["merge_path"] = {
                "bytes_8d_per_step": bytes_8d, "ms_per_step": round(path_ms, 4),
                "achieved_8d": round(bytes_8d / (path_ms / 1e3) / 1e9, 1), "unit": "GB/s",
                "frac_8d": round(bytes_8d / (path_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                "launches_per_layer": (4 if filt else 3) + 1,
                "launches": ("k_unit_rows_f, k_scores_filter, k_exact_rows (+ fp32 fallback waves), k_rank_select"
                             if filt else "k_unit_rows_heads, k_scores_rowmax, k_rank_select") + ", k_merge_rows_fast<LN>",
                "note": "SURVEY 8d bytes of the 12 merge calls / time of all their launches (matching stages timed inside "
                        "tome_match_keys, the merge kernel inside the forward); call counts: profiles/r04_*kernel_stats*",
                "matching_in_forward": matching_placement(B, t0_tokens)}
            from tome import _overlap
            if _overlap.ENABLED and B * t0_tokens * t0_tokens >= _overlap.MIN_WORK:
                # what the path costs the STEP once the matching runs beside the attention: the merge launches alone
                # (profiles/r04_videomae_forward_composition.txt: the step is the sum of every kernel but the matching's)
                crit_ms = stats["k_merge_rows"]["ms"]
                out["merge_path"]["on_the_critical_path"] = {
                    "launches": "k_merge_rows_fast<LN>", "ms_per_step": round(crit_ms, 4),
                    "achieved_8d": round(bytes_8d / (crit_ms / 1e3) / 1e9, 1),
                    "frac_8d": round(bytes_8d / (crit_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4)}
            with torch.no_grad():
                out["attention_kernel"] = measure_attention(B, t0_tokens, args.r, dev)
        if world == 1 and not args.no_also:
            del model, clips
            torch.cuda.empty_cache()
            try:
                out["also"] = also_workloads(dev, quick=args.also_quick)
            except Exception as exc:  # the headline line must not be lost to a secondary workload
                out["also"] = None
                out["also_error"] = f"{type(exc).__name__}: {exc}"[:500]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.frames, args.r, args.cpu_clips, args.cpu_iters)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def main():
    """`python bench.py --gpus N`: under torch.distributed.run the ranks exist already (WORLD_SIZE must equal N);
    started plainly with N > 1 the program spawns its N ranks itself -- fresh interpreters over a TCP rendezvous
    on 127.0.0.1, as the reference's launch_job does (slowfast/utils/misc.py:402-430) -- before this process has
    issued any GPU call, and only waits for them."""
    from hosts import launch
    args = parse()
    if launch.under_launcher():
        launch.check_world(args.gpus)
    launch.run(worker, args.gpus, (args,))


if __name__ == "__main__":
    main()
