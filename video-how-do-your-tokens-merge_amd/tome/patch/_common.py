"""Shared machinery of the four model patches (reference: tome/patch/{videomae,timesformer,motionformer,
vivit}.py).  The reference imports the concrete slowfast / HF classes and swaps ``__class__``; here the
ToMe subclasses are created on the fly from whatever class the module already has, so the same patch
works on the reference's own model objects and on the host models of this repository
(``video-how-do-your-tokens-merge_amd/hosts``) without importing slowfast.
"""
from __future__ import annotations

import os
import threading
from typing import Callable, Dict, Tuple

import torch

from .. import _overlap
from ..merge import (bipartite_soft_matching, bipartite_soft_matching_drop, bipartite_soft_matching_hybrid,
                     merge_source, merge_wavg)
from ..utils import parse_r

_SUBCLASSES: Dict[Tuple[type, str], type] = {}
_FUSE_LN = os.environ.get("TOME_FUSE_LN", "1") != "0"  # measurement switch: 0 = merge and LayerNorm as two steps
_FUSE_ADD = os.environ.get("TOME_FUSE_ADD", "1") != "0"  # measurement switch: 0 = residual add as its own pass
_FUSE_NEXT = os.environ.get("TOME_FUSE_NEXT", "1") != "0"  # 0 = second residual and the next block's norm1 separate
_ATTN_KERNEL = os.environ.get("TOME_ATTN_KERNEL", "1") != "0"  # 0 = the framework's fused attention (+ bias tensor)
_GELU_KERNEL = os.environ.get("TOME_GELU_KERNEL", "1") != "0"  # 0 = the framework's GELU pass inside the MLP
_FUSE_FC2 = os.environ.get("TOME_FUSE_FC2", "1") != "0"  # 0 = the MLP's second GEMM writes its own tensor, then an add
_SKIP_FIRST = os.environ.get("TOME_SKIP_FIRST", "1") != "0"  # the temporal hand-over without the class row


def swizzle(module: torch.nn.Module, tag: str, methods: dict) -> None:
    """Give ``module`` a subclass of its current class carrying ``methods`` (the reference assigns
    ``module.__class__ = ToMeBlock``; same effect, no dependency on the concrete base class)."""
    base = module.__class__
    if getattr(base, "_tome_tag", None) == tag:
        return
    key = (base, tag)
    sub = _SUBCLASSES.get(key)
    if sub is None:
        sub = type(tag, (base,), dict(methods, _tome_tag=tag))
        _SUBCLASSES[key] = sub
    module.__class__ = sub


def has_tag(module, tag: str) -> bool:
    return getattr(module.__class__, "_tome_tag", None) == tag


def new_tome_info(trace_source, prop_attn, mode, head_aggregation, threshold, verbose, class_token) -> dict:
    """The shared per-model state dict (videomae.py:181-193)."""
    return {
        "r": 0,
        "size": None,
        "source": None,
        "trace_source": trace_source,
        "prop_attn": prop_attn,
        "verbose": verbose,
        "class_token": class_token,
        "distill_token": False,
        "mode": mode,
        "head_aggregation": head_aggregation,
        "threshold": threshold,
    }


# One forward in flight per process.  Two forwards (patched or not) issued on two HIP streams of one process never
# finish on this platform: two chains of plain library GEMMs (`torch.mm`, nothing of this package) on two streams are
# not finished after 20 s where one stream takes 88 ms (tools/probes/two_stream_gemm.py,
# profiles/r03_two_stream_gemm_probe.txt).  Every GEMM of these models -- under either BLAS preference, as the kernel
# trace in profiles/r04_two_stream_probe_rocblas_kernel.txt shows -- is a persistent Stream-K grid (`..._SK3_...`, one
# workgroup per CU, workgroups spin on each other's partial tiles); two such grids resident at once waiting on
# siblings that cannot be scheduled is the likely cause, but the probe has no non-Stream-K control, so the rule below
# is stated for "two concurrent library GEMM grids".
# The reference's own contract is one forward at a time per model (`_tome_info` is shared state, SURVEY 8b
# "Threading").  So: (1) the ENQUEUEING of a patched forward holds a per-device lock -- a second host thread that
# starts a forward while the first is still issuing kernels (launches release the GIL) waits at the entry, and finds
# the first one's end event when it gets in; (2) a patched forward issued while another one is still in flight on a
# different stream is ORDERED behind it (the stream waits for the other forward's end event; said once in a warning)
# instead of hanging the device: two streams then buy no overlap, and nothing deadlocks.  TOME_ONE_FORWARD=raise
# refuses instead.
_in_flight = {}  # device index -> (stream id, event recorded behind the last patched forward)
_issue_locks: Dict[int, "threading.RLock"] = {}
_issue_locks_guard = threading.Lock()
_warned_two_streams = False


def _current_stream(device):
    """(stream id, stream) of the caller's current HIP stream -- a seam for the CPU-side test of the bookkeeping."""
    cur = torch.cuda.current_stream(device)
    return cur.cuda_stream, cur


def _record_event(stream):
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev


def _guarded(device) -> bool:
    return device.type == "cuda" and not torch.cuda.is_current_stream_capturing()
    # (a captured forward is ordered by the graph it is replayed from)


def _issue_lock(device):
    with _issue_locks_guard:
        lock = _issue_locks.get(device.index)
        if lock is None:
            lock = _issue_locks[device.index] = threading.RLock()
        return lock


def _guard_one_forward_in_flight(device) -> None:
    """Called at the ENTRY of a patched forward, with the device's issue lock held."""
    global _warned_two_streams
    sid, cur = _current_stream(device)
    last = _in_flight.get(device.index)
    if last is None or last[0] == sid or last[1].query():
        return
    msg = ("a forward issued on another HIP stream of this process is still in flight: the library GEMMs of two "
           "concurrent forwards never finish on this platform")
    if os.environ.get("TOME_ONE_FORWARD", "order") == "raise":
        raise RuntimeError(msg + ".  Run one forward at a time per process (one process per GPU).")
    cur.wait_event(last[1])
    if not _warned_two_streams:
        _warned_two_streams = True
        import warnings
        warnings.warn(msg + "; this forward has been ordered behind it (no overlap between the two streams).",
                      RuntimeWarning, stacklevel=3)


def _note_forward_issued(device) -> None:
    sid, cur = _current_stream(device)
    _in_flight[device.index] = (sid, _record_event(cur))


class one_forward_at_a_time:
    """Context of one patched forward on `device`: takes the device's issue lock (re-entrant: a patched model called
    from inside another patched forward of the same thread), orders the caller's stream behind a forward still in
    flight on another stream, and records the end event on exit -- also when the forward raises, since whatever it
    enqueued before raising is in flight all the same."""

    def __init__(self, device):
        self.device = device
        self.on = device is not None and _guarded(device)

    def __enter__(self):
        if self.on:
            self.lock = _issue_lock(self.device)
            self.lock.acquire()
            try:
                _guard_one_forward_in_flight(self.device)
            except BaseException:
                self.lock.release()
                raise
        return self

    def __exit__(self, *exc):
        if self.on:
            try:
                _note_forward_issued(self.device)
            finally:
                self.lock.release()
        return False


def wrap_model_forward(model_wrapper: torch.nn.Module, blocks_of: Callable) -> None:
    """make_tome_class (videomae.py:160-169): every forward re-reads ``self.r`` into a per-layer list and
    clears size/source."""
    base = model_wrapper.__class__
    if getattr(base, "_tome_tag", None) == "ToMeVisionTransformer":
        return

    def forward(self, *args, **kwdargs):
        param = next(self.parameters(), None)
        with one_forward_at_a_time(None if param is None else param.device):
            # (the shared state is reset under the lock: `_tome_info` belongs to the forward that holds it)
            self._tome_info["r"] = parse_r(len(blocks_of(self)), self.r)
            self._tome_info["size"] = None
            self._tome_info["source"] = None
            self._tome_info.pop("_prenorm", None)
            self._tome_info.pop("_folded", None)
            try:
                return super(sub, self).forward(*args, **kwdargs)
            finally:
                if param is not None:
                    _overlap.join(param.device)  # (a matching on the side stream no block waited for: a raise mid-block)

    sub = type("ToMeVisionTransformer", (base,), {"forward": forward, "_tome_tag": "ToMeVisionTransformer"})
    model_wrapper.__class__ = sub


keys_ready = _overlap.keys_ready      # the block's matching beside its attention, on a second HIP stream:
match_beside = _overlap.match_beside  # tome/_overlap.py


def attention(q, k, v, size, scale: float, dropout_p: float = 0.0, bias_skip: bool = False):
    """softmax(q k^T * scale + log(size)) v for [B, H, N, hd] head views; returns [B, N, H*hd].
    16-bit heads of width 64 without dropout go through tome_prop_attention (the size bias is one value per key
    inside the kernel, q/k/v are read in place from the projection's output); everything else through the
    framework's attention with the bias tensor the reference builds (videomae.py:62-63, timesformer.py:73-74)."""
    from .. import _abi
    B, H, N, hd = q.shape
    if (_ATTN_KERNEL and dropout_p == 0.0 and _abi.prop_attention_ok(q) and _abi.prop_attention_ok(k)
            and _abi.prop_attention_ok(v)):
        return _abi.prop_attention(q, k, v, size, scale, bias_skip=bias_skip, checked=True)
    bias = None
    if size is not None:
        log = _abi.log_of_size(size)[:, None, None, :, 0].to(q.dtype)
        if bias_skip:
            bias = torch.zeros(B, 1, N, N, dtype=q.dtype, device=q.device)
            bias[:, :, 1:, 1:] = log
        else:
            bias = log
    out = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=bias, dropout_p=dropout_p, scale=scale)
    return out.transpose(1, 2).reshape(B, N, H * hd)


def _stock_module(m, cls) -> bool:
    """`m` is exactly `cls` (not a subclass with a forward of its own: LoRA, quantised, ... layers), carries no
    parametrization and no forward hook -- only then may its forward be replaced by a hand-made call."""
    return (type(m) is cls and not getattr(m, "parametrizations", None)
            and not m._forward_hooks and not m._forward_pre_hooks)


def _plain_mlp(mlp) -> bool:
    """An MLP of the usual shape: fc1, exact-erf nn.GELU, fc2, dropouts that are the identity in eval -- all of them
    the stock modules, unhooked (feature extractors / flop counters hook mlp, act, fc2: those run the module itself)."""
    act = getattr(mlp, "act", None)
    fc1, fc2 = getattr(mlp, "fc1", None), getattr(mlp, "fc2", None)
    return (act is not None and _stock_module(act, torch.nn.GELU) and getattr(act, "approximate", "none") == "none"
            and fc1 is not None and _stock_module(fc1, torch.nn.Linear)
            and fc2 is not None and _stock_module(fc2, torch.nn.Linear) and not mlp.training
            and not mlp._forward_hooks and not mlp._forward_pre_hooks
            and set(dict(mlp.named_children())) <= {"fc1", "act", "fc2", "drop", "drop1", "drop2"})


def mlp_hidden(mlp, y):
    """fc1 and the activation of a plain MLP: the tensor its fc2 reads.  The activation runs on tome_gelu_erf (same
    bits as the framework's kernel, non-temporal streaming: 394 -> ~350 us at batch 128)."""
    from .. import _abi
    h = mlp.fc1(y)
    if _GELU_KERNEL and _abi.gelu_ok(h):
        return _abi.gelu_erf(h, inplace=True)
    return mlp.act(h)


def run_mlp(mlp, y):
    """`self.mlp(y)` of the patched block (tome/patch/videomae.py:29); anything but a plain MLP is called as it is."""
    if _plain_mlp(mlp):
        return mlp.fc2(mlp_hidden(mlp, y))
    return mlp(y)


def foldable(linear, eval_mode: bool = True):
    """`linear` when the block's last step `x = x + linear(h)` may run as ONE GEMM that accumulates onto the residual
    stream in place (`x.addmm_(h, Wᵀ)`, finish_linear) -- which needs the bias in the stream beforehand
    (merge_then_norm's `fold`, the merge kernel's x_out_bias); None when it may not."""
    if (_FUSE_FC2 and _FUSE_NEXT and eval_mode and _stock_module(linear, torch.nn.Linear) and linear.bias is not None
            and not (torch.is_grad_enabled() and linear.weight.requires_grad)):
        return linear
    return None


def mlp_residual(block, mlp, x, y, info, scale=None, drop_path=None):
    """`x + drop_path(scale * mlp(y))` at the end of a patched block (tome/patch/videomae.py:28-29,
    timesformer.py:57, motionformer.py:30), with the next block's first norm handed over (finish_block)."""
    if _plain_mlp(mlp) and scale is None and (drop_path is None or not block.training):
        return finish_linear(block, x, mlp_hidden(mlp, y), mlp.fc2, info)
    if info.get("_folded") is not None and info["_folded"][0] is x:
        raise RuntimeError("the residual stream carries a folded bias, but the block's MLP is not the one it was folded for")
    y = run_mlp(mlp, y)
    if scale is not None:
        y = scale * y
    return finish_block(block, x, y if drop_path is None else drop_path(y), info)


def pick_reduction(mode: str, merge_fn, drop_fn, hybrid_fn):
    if mode in ("merge", "random_merge"):
        return merge_fn
    if mode in ("drop", "random_drop"):
        return drop_fn
    if mode in ("hybrid",):
        return hybrid_fn
    raise ValueError(f"unknown ToMe mode {mode!r}")


# ---- the reduction step on an already grouped [n, T, C] token tensor ---------------------------------
def reduce_merge(metric, x, info, r):
    merge, _ = bipartite_soft_matching(metric, r, info["class_token"], info["distill_token"], info["mode"])
    if info["trace_source"]:
        info["source"] = merge_source(merge, x, info["source"])
    before = x.size(1)
    x, info["size"] = merge_wavg(merge, x, info["size"], log_size=info["prop_attn"])
    if info["verbose"]:
        print(f"Merged {before} to {x.size(1)} tokens")
    return x


def link_next_norms(blocks, norm_attr: str, tag: str = "ToMeBlock", skip_first: bool = False) -> None:
    """Tell every patched block which LayerNorm reads its output (the next block's first norm), so the block's
    last residual add can hand that norm's result over (finish_block / first_norm).  Stored without registering
    the norm as a submodule of the previous block.  skip_first: that norm's consumer reads `norm(x)[:, 1:]` only
    (TimeSformer's temporal_norm1), so the hand-over leaves the class-token row out."""
    blocks = list(blocks)
    for cur, nxt in zip(blocks, blocks[1:]):
        target = getattr(nxt, norm_attr, None) if has_tag(nxt, tag) and nxt is not cur else None
        object.__setattr__(cur, "_tome_next_norm", target)
        object.__setattr__(cur, "_tome_next_skip_first", bool(skip_first) and _SKIP_FIRST)
    if blocks:
        object.__setattr__(blocks[-1], "_tome_next_norm", None)


def first_norm(block, x, info, norm, skip_first: bool = False):
    """norm(x) at the top of a block -- taken from the previous block's fused add+LayerNorm when it left one
    for exactly this tensor.  skip_first: returns norm(x)[:, 1:] (contiguous when it comes from the hand-over)."""
    pre = info.pop("_prenorm", None)
    if pre is not None and pre[0] is x and pre[2] is norm:
        if bool(pre[3]) == bool(skip_first):
            return pre[1]
        if skip_first:  # a full hand-over for a reader of the patch rows
            return pre[1][:, 1:, :]
    from .. import _abi
    if _FUSE_NEXT and isinstance(norm, torch.nn.LayerNorm) and _abi.ln_fusable(x, norm):
        # no hand-over (first block, or a block after one that could not fuse): the same streaming LayerNorm kernel,
        # without an addend (6.4 TB/s against 2 TB/s for the framework's LayerNorm on these shapes)
        skip = skip_first and x.dim() == 3 and x.shape[1] >= 2
        y = _abi.add_layernorm(x, None, norm.weight, norm.bias, norm.eps, skip_first=skip)[1]
        return y[:, 1:, :] if (skip_first and not skip) else y
    return norm(x)[:, 1:, :] if skip_first else norm(x)


def finish_block(block, x, residual, info):
    """x + residual at the end of a block; when the next block's first LayerNorm is known and the tokens are
    16-bit, tome_add_layernorm produces the sum and that norm's output together."""
    from .. import _abi
    nxt = getattr(block, "_tome_next_norm", None)
    if _FUSE_NEXT and nxt is not None and residual.dtype == x.dtype and _abi.ln_fusable(x, nxt):
        skip = bool(getattr(block, "_tome_next_skip_first", False)) and x.dim() == 3 and x.shape[1] >= 2
        x, h = _abi.add_layernorm(x, residual, nxt.weight, nxt.bias, nxt.eps, skip_first=skip)
        info["_prenorm"] = (x, h, nxt, skip)
        return x
    return x + residual


def finish_linear(block, x, h, linear, info):
    """`x + linear(h)` at the end of a block.  When the merge kernel has put linear's bias into x already (`fold`),
    the GEMM accumulates onto x in place (beta = 1: no tensor for the GEMM's result, no add pass) and the next block's
    first LayerNorm reads the finished sum once (tome_add_layernorm without addend); otherwise finish_block."""
    from .. import _abi
    folded = info.pop("_folded", None)
    if folded is None or folded[0] is not x:
        return finish_block(block, x, linear(h), info)
    if folded[1] is not linear:
        raise RuntimeError("the residual stream carries the bias of another Linear than the one that finishes the block")
    x.view(-1, x.shape[-1]).addmm_(h.view(-1, h.shape[-1]), linear.weight.t())
    nxt = getattr(block, "_tome_next_norm", None)
    if nxt is not None and _abi.ln_fusable(x, nxt):
        skip = bool(getattr(block, "_tome_next_skip_first", False)) and x.dim() == 3 and x.shape[1] >= 2
        _, hn = _abi.add_layernorm(x, None, nxt.weight, nxt.bias, nxt.eps, skip_first=skip)
        info["_prenorm"] = (x, hn, nxt, skip)
    return x


def merge_then_norm(metric, x, info, norm, reduction_function, plain_merge_fn, residual=None, fold=None):
    """The block steps `[x = x + residual;] x = reduction_function(metric, x, info); y = norm(x)` with the
    residual add and the LayerNorm fused into the merge kernel (tome_merge_wavg_ln) when this layer merges in
    plain 'merge' mode on 16-bit tokens; returns (x, y).  Anything else runs the steps as the reference does.
    fold: the Linear that finishes the block (`foldable`); when the fused kernel runs, x comes back with that bias
    added (y is the norm of x without it) and info["_folded"] says so for finish_linear."""
    from .. import _abi
    from ..merge import do_nothing
    r_list = info["r"]
    if residual is not None and not (_FUSE_LN and _FUSE_ADD and reduction_function is plain_merge_fn and r_list
                                     and r_list[0] > 0 and info["mode"] == "merge" and not info["trace_source"]
                                     and _abi.ln_fusable(x, norm) and residual.dtype == x.dtype
                                     and _abi.effective_r(x.shape[1], r_list[0], info["class_token"],
                                                          info["distill_token"]) > 0):
        x = x + residual
        residual = None
    if (_FUSE_LN and reduction_function is plain_merge_fn and r_list and r_list[0] > 0 and info["mode"] == "merge"
            and _abi.ln_fusable(x, norm)):
        r = r_list.pop(0)
        merge, _ = bipartite_soft_matching(metric, r, info["class_token"], info["distill_token"], info["mode"])
        if merge is do_nothing:
            if residual is not None:
                x = x + residual
            return x, norm(x)
        if info["trace_source"]:
            info["source"] = merge_source(merge, x, info["source"])
        before = x.size(1)
        fold = fold if (fold is not None and fold.bias.dtype == x.dtype and fold.bias.numel() == x.shape[-1]) else None
        x, y, info["size"] = _abi.merge_wavg_ln(merge.plan, x, info["size"], norm.weight, norm.bias, norm.eps,
                                                addend=residual, log_size=info["prop_attn"],
                                                out_bias=None if fold is None else fold.bias)
        if fold is not None:
            info["_folded"] = (x, fold)
        if info["verbose"]:
            print(f"Merged {before} to {x.size(1)} tokens")
        return x, y
    x = reduction_function(metric, x, info)
    return x, norm(x)


def _regrouped_by_views(reduce_grouped, metric, x_full, info, r, frames):
    """The reference's own sequence around a grouped reduction (timesformer.py:89-107, motionformer.py:150-168): split
    the class token off, '(p t) -> (b t) p', reduce, back, cat.  Differentiable framework ops end to end: the form the
    regrouped reductions take when the tokens require grad (training, tools/train_net.py:727-741)."""
    B, N, C = x_full.shape
    P = (N - 1) // frames
    body = x_full[:, 1:].reshape(B, P, frames, C).transpose(1, 2).reshape(B * frames, P, C)
    body = reduce_grouped(metric, body, info, r)
    P2 = body.shape[1]
    body = body.reshape(B, frames, P2, C).transpose(1, 2).reshape(B, P2 * frames, C)
    return torch.cat((x_full[:, :1], body), dim=1)


def _training_tokens(x) -> bool:
    return torch.is_grad_enabled() and x.requires_grad


def reduce_merge_regrouped(metric, x_full, info, r, frames, hybrid=False):
    """reduce_merge / reduce_hybrid for the models whose merge groups are interleaved in the token sequence
    (TimeSformer '(p t)', Motionformer '(s f)'): x_full is [B, 1 + P*F, C] with the class token in front; the
    kernel addresses the groups in place and returns [B, 1 + (P-r)*F, C] (no permuted copies of x)."""
    from .. import _abi
    from ..merge import do_nothing
    if _training_tokens(x_full):
        return _regrouped_by_views(reduce_hybrid if hybrid else reduce_merge, metric, x_full, info, r, frames)
    if hybrid:
        merge, _ = bipartite_soft_matching_hybrid(metric, r, info["class_token"], info["distill_token"], info["mode"],
                                                  info["threshold"])
    else:
        merge, _ = bipartite_soft_matching(metric, r, info["class_token"], info["distill_token"], info["mode"])
    if merge is do_nothing:
        return x_full
    plan = merge.plan
    if info["trace_source"]:
        shape_only = x_full.new_empty((plan.n, plan.T, 0))
        info["source"] = merge_source(merge, shape_only, info["source"])
    x_out, info["size"] = _abi.merge_wavg_regrouped(plan, x_full, info["size"], frames, has_cls=True,
                                                    log_size=info["prop_attn"])
    if info["verbose"]:
        print(f"Merged {plan.T} to {plan.T - plan.r} tokens")
    return x_out


class GroupedResidual:
    """TimeSformer's second residual as the spatial attention leaves it (tome/patch/timesformer.py:32-52): `rs`
    [(b t), 1 + p, m] with a class row per frame, and the class token averaged over the frames `cls_new` [b, 1, m].
    The reference rearranges it '(b t) p m -> b (p t) m' and concatenates; the fused merge kernel reads it where it
    lies (`addend_grouped`), anything else asks for `.materialize()`."""

    def __init__(self, rs: torch.Tensor, cls_new: torch.Tensor, B: int, T: int, P: int):
        self.rs, self.cls_new, self.B, self.T, self.P = rs, cls_new, B, T, P
        self.dtype = rs.dtype

    def materialize(self) -> torch.Tensor:
        B, T, P, m = self.B, self.T, self.P, self.rs.shape[-1]
        body = self.rs.reshape(B, T, 1 + P, m)[:, :, 1:, :].transpose(1, 2)  # '(b t) p -> b p t', still a view
        if torch.is_grad_enabled() and self.rs.requires_grad:
            return torch.cat((self.cls_new, body.reshape(B, P * T, m)), 1)
        res = torch.empty((B, 1 + P * T, m), dtype=self.rs.dtype, device=self.rs.device)
        res[:, :1, :] = self.cls_new  # assembled by one strided copy
        res[:, 1:, :].view(B, P, T, m).copy_(body)
        return res


def merge_then_norm_regrouped(metric, x_full, info, norm, unfused_reduce, is_plain_merge: bool, frames: int,
                              residual=None, fold=None):
    """merge_then_norm for the interleaved layouts (TimeSformer '(p t)', Motionformer '(s f)'): x_full is
    [B, 1 + P*F, C]; `unfused_reduce(x)` is the model's own reduction step (it pops r itself).  `residual`: a tensor
    in x's layout, or a GroupedResidual."""
    from .. import _abi
    from ..merge import do_nothing
    r_list = info["r"]
    fusable = (_FUSE_LN and is_plain_merge and r_list and r_list[0] > 0 and info["mode"] == "merge"
               and not info["trace_source"] and _abi.ln_fusable(x_full, norm)
               and _abi.effective_r((x_full.shape[1] - 1) // frames, r_list[0], False, False) > 0)
    grouped = residual if isinstance(residual, GroupedResidual) else None
    if grouped is not None and not (fusable and _FUSE_ADD and grouped.dtype == x_full.dtype):
        residual, grouped = grouped.materialize(), None
    if not fusable:
        if residual is not None:
            x_full = x_full + residual
        x_full = unfused_reduce(x_full)
        return x_full, norm(x_full)
    if grouped is None and residual is not None and (not _FUSE_ADD or residual.dtype != x_full.dtype):
        x_full = x_full + residual
        residual = None
    r = r_list.pop(0)
    merge, _ = bipartite_soft_matching(metric, r, info["class_token"], info["distill_token"], info["mode"])
    assert merge is not do_nothing
    plan = merge.plan
    fold = fold if (fold is not None and fold.bias.dtype == x_full.dtype
                    and fold.bias.numel() == x_full.shape[-1]) else None
    out_bias = None if fold is None else fold.bias
    if grouped is not None:
        x_out, y_out, info["size"] = _abi.merge_wavg_regrouped(
            plan, x_full, info["size"], frames, has_cls=True, ln=(norm.weight, norm.bias, norm.eps),
            addend_grouped=grouped.rs, cls_addend=grouped.cls_new, log_size=info["prop_attn"], out_bias=out_bias)
    else:
        x_out, y_out, info["size"] = _abi.merge_wavg_regrouped(plan, x_full, info["size"], frames, has_cls=True,
                                                              ln=(norm.weight, norm.bias, norm.eps), addend=residual,
                                                              log_size=info["prop_attn"], out_bias=out_bias)
    if fold is not None:
        info["_folded"] = (x_out, fold)
    if info["verbose"]:
        print(f"Merged {plan.T} to {plan.T - plan.r} tokens")
    return x_out, y_out


def _drop_source(drop, source, n, t, device):
    """`drop(source)` with `source = eye(T)` when None (tome/patch/videomae.py:112-117); the first layer's matrix
    comes straight from the row map (tome_source_init), without the identity."""
    from .. import _abi
    plan = getattr(drop, "plan", None)
    if source is None:
        if plan is not None:
            return _abi.source_init(plan, drop=True)
        source = torch.eye(t, device=device)[None, ...].expand(n, t, t)  # nothing to drop (r clamped to 0)
    return drop(source)


def reduce_drop(metric, x, info, r):
    drop = bipartite_soft_matching_drop(metric, r, info["class_token"], info["distill_token"], info["mode"])
    if isinstance(drop, tuple):  # clamped r == 0: the reference returns the do_nothing pair here
        drop = drop[0]
    if info["trace_source"]:
        info["source"] = _drop_source(drop, info["source"], x.shape[0], x.shape[1], x.device)
    before = x.size(1)
    x = drop(x)
    info["size"] = torch.ones((x.size(0), x.size(1), 1), device=x.device)
    if info["verbose"]:
        print(f"Dropped {before} to {x.size(1)} tokens")
    return x


def reduce_drop_regrouped(metric, x_full, info, r, frames: int):
    """reduce_drop for the interleaved layouts: x_full [B, 1 + P*F, C] with the class token in front; the groups
    are addressed in place by the kernel (tome_drop_regrouped) instead of regrouped by permuted copies
    (timesformer.py:111-131, motionformer.py:172-193)."""
    from .. import _abi
    if _training_tokens(x_full):
        return _regrouped_by_views(reduce_drop, metric, x_full, info, r, frames)
    drop = bipartite_soft_matching_drop(metric, r, info["class_token"], info["distill_token"], info["mode"])
    if isinstance(drop, tuple):
        return x_full
    plan = drop.plan
    if info["trace_source"]:
        info["source"] = _drop_source(drop, info["source"], plan.n, plan.T, x_full.device)
    x_out = _abi.drop_regrouped(plan, x_full, frames, has_cls=True)
    info["size"] = torch.ones((plan.n, plan.T - plan.r, 1), device=x_full.device)
    if info["verbose"]:
        print(f"Dropped {plan.T} to {plan.T - plan.r} tokens")
    return x_out


def reduce_hybrid(metric, x, info, r):
    merge, _ = bipartite_soft_matching_hybrid(metric, r, info["class_token"], info["distill_token"], info["mode"],
                                              info["threshold"])
    if info["trace_source"]:
        info["source"] = merge_source(merge, x, info["source"])
    before = x.size(1)
    x, info["size"] = merge_wavg(merge, x, info["size"], log_size=info["prop_attn"])
    if info["verbose"]:
        print(f"Merged {before} to {x.size(1)} tokens")
    return x
