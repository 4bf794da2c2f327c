"""The block's matching on a second HIP stream, beside the block's attention.

The reference runs a block as `qkv GEMM -> attention -> proj GEMM -> bipartite_soft_matching(k.mean(1)) -> merge`
(`tome/patch/videomae.py:14-30, 48-77`), one op after the other.  The matching needs nothing but the keys, and those
exist as soon as the qkv GEMM is done; the attention kernel that follows is bound by the matrix and vector pipes with HBM
nearly idle and leaves a quarter of the register file free, and the matching's largest kernel (`k_unit_rows_f`: all heads'
keys once, 28 registers) is bound by HBM.  So a patched attention marks the point behind its qkv GEMM (`keys_ready`),
issues its attention kernel, and hands the keys to `match_beside`, which runs `tome_match_keys` on a per-device side
stream behind that mark; `bipartite_soft_matching` finds the plan on the metric (`HeadMeanKeys.early`), makes the
caller's stream wait for the side stream, and uses it.  Same kernels, same inputs, same plan: nothing about the result
depends on the stream.  Measured (`tools/probes/overlap_match_attention.py`, `tools/overlap_ab.py`,
`profiles/r04_overlap_probe.txt`): attention + projection + matching 1544 -> 1447 us at batch 128; whole forward +1.0..1.6 %
on VideoMAE, +0.8..4.1 % on ViViT.  TimeSformer / Motionformer fork only inside a HIP-graph capture (+0.3..1.3 %): their resident
attention kernel fills the register file (nothing runs beside it: +-0 for eager forwards at large batches) and at batch 8
they are bound by the host, where the stream and event calls of a fork cost 15-20 %.

Why this cannot deadlock where two concurrent forwards do (`tome/patch/_common.py`, "One forward in flight"): only this
package's matching kernels ever run on the side stream -- no library GEMM -- and none of them waits for another
workgroup, so at most one persistent Stream-K grid is resident at any time.

Memory: plan and scratch are allocated while the side stream is the CURRENT stream, i.e. from the caching allocator's
pool of that stream.  A block of that pool has only ever been touched by side-stream work and by the main stream's
reads of a plan -- and the side stream always starts behind an event of the main stream recorded after those reads,
while the main stream always waits for the side stream (`join`; the patched model forward joins once more on exit, also
when a block raised) before it reads the plan; the keys are held by the metric until then.  (Round 4 tried the cheaper
form -- launch on the side stream by raw handle, allocate from the MAIN stream's pool: wrong.  That pool hands out blocks
whose last main-stream reader may still be running -- a temporary the attention wrapper released a microsecond earlier
-- and the matching then writes into them beside that reader: TimeSformer's attention output changed at batch 64,
`tools/probes/overlap_diverge.py`; `test_side_stream_matching_shares_no_memory_with_kernels_in_flight` holds the line.)
Two events per device, re-recorded every layer (a wait refers to the record that precedes it).  Under HIP-graph capture
the side stream becomes part of the capture through the same two events.

`TOME_MATCH_STREAM=0` keeps the matching on the caller's stream (measurement switch); `TOME_MATCH_STREAM_MIN` is the
size from which an eager forward forks (below)."""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

ENABLED = os.environ.get("TOME_MATCH_STREAM", "1") != "0"
# Eager launches that alternate between two streams cost the host 30-50 us per layer (measured:
# profiles/r04_overlap_probe.txt, "host issue"); a forward whose device time is close to its host time -- VideoMAE at the
# reference's batch of 8: 4.7 ms against 3.1-3.9 ms -- gains 2 % on a quiet host and loses 15 % on a busy one.  So the
# fork needs groups x tokens^2 (what the attention beside it scales with) of at least this, or a HIP-graph capture
# (no host in the replay): VideoMAE from batch 32, ViViT-3137 from batch 8.
MIN_WORK = int(float(os.environ.get("TOME_MATCH_STREAM_MIN", "6e7")))
_side: Dict[int, tuple] = {}  # device index -> (side stream, its raw handle, fork event, join event)
_open: Dict[int, tuple] = {}  # device index -> the same, while the side stream has work the main stream has not waited for


def _index(device) -> int:
    return torch.cuda.current_device() if device.index is None else device.index


def _state(device) -> tuple:
    idx = _index(device)
    st = _side.get(idx)
    if st is None:
        s = torch.cuda.Stream(device=device)
        st = _side[idx] = (s, s.cuda_stream, torch.cuda.Event(), torch.cuda.Event())
    return st


def side_stream(device) -> "torch.cuda.Stream":
    return _state(device)[0]


def keys_ready(keys: torch.Tensor, info: Optional[dict], capture_only: bool = False):
    """Between a patched attention's qkv GEMM and its attention kernel: the event on the caller's stream behind which
    `keys` exist -- or None when this layer's matching stays on the caller's stream (switch off, CPU tensors, no plain
    merge this layer, keys the kernel cannot read in place).  capture_only: fork only inside a HIP-graph capture
    (TimeSformer / Motionformer: an eager fork gains nothing at large batches and costs the host at small ones)."""
    if not ENABLED or info is None or not keys.is_cuda or info.get("mode") != "merge":
        return None
    if capture_only and not torch.cuda.is_current_stream_capturing():
        return None
    r_list = info.get("r")
    if not isinstance(r_list, list) or not r_list or r_list[0] <= 0:
        return None
    from . import _abi
    tokens = keys.shape[-2]
    if _abi.effective_r(tokens, r_list[0], info["class_token"], info["distill_token"]) <= 0 \
            or not _abi.keys_fusable(keys):
        return None
    if keys.shape[0] * tokens * tokens < MIN_WORK and not torch.cuda.is_current_stream_capturing():
        return None
    ev = _state(keys.device)[2]
    ev.record(torch.cuda.current_stream(keys.device))
    return ev


def match_beside(metric, ready, info: dict) -> None:
    """Run this layer's matching of `metric` (a HeadMeanKeys) on the side stream behind `ready` (from `keys_ready`);
    the plan is left on the metric for `tome.merge` to pick up."""
    if ready is None:
        return
    from . import _abi
    dev = metric.keys.device
    st = _state(dev)
    st[0].wait_event(ready)
    r, cls, dist = int(info["r"][0]), bool(info["class_token"]), bool(info["distill_token"])
    _open[_index(dev)] = st
    with torch.cuda.stream(st[0]):  # launches AND allocations on the side stream (see "Memory" above)
        plan = _abi.match_keys(metric.keys, r, cls, dist, checked=True)
    metric.early = (r, cls, dist, plan)


def join(device) -> None:
    """The caller's current stream waits for whatever the side stream of `device` still has in flight."""
    if device.type != "cuda":
        return
    st = _open.pop(_index(device), None)
    if st is not None:
        st[3].record(st[0])
        torch.cuda.current_stream(device).wait_event(st[3])


def take(metric, r, class_token, distill_token):
    """(found, plan) -- the plan `match_beside` left on `metric`, after making the caller's stream wait for it.  A
    plan made for other arguments is dropped (found False); the wait happens all the same."""
    early = getattr(metric, "early", None)
    if early is None:
        return False, None
    metric.early = None
    join(metric.keys.device)
    if early[:3] == (int(r), bool(class_token), bool(distill_token)):
        return True, early[3]
    return False, None
